"""Parity of the HIP prune/regrow kernels (K2) with the oracle and with the
golden vectors produced by the reference's own code.  Bit-exact: masks,
weights and momentum are compared as integers."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
from oracle import rigl_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


def _dev():
  return torch.device('cuda:0')


def _t(a):
  return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32).reshape(-1)).to(_dev())


def _run_rigl(ops, mask, w, g, frac, noise=None, mom=None, grow_init='zeros',
              acc_scale=0.0):
  from rigl_amd import _lib
  n = mask.size
  tw, tg = _t(w), _t(g)
  tm = _t(mom) if mom is not None else None
  tn = _t(noise) if noise is not None else None
  bits = ops.mask_pack(_t(mask))
  mode, div = _lib.GROW_ZEROS, 1.0
  if grow_init.startswith('grad_scale'):
    mode, div = _lib.GROW_GRAD_SCALE, O.extract_number(grow_init)
  elif grow_init.startswith('grad_sign'):
    mode, div = _lib.GROW_GRAD_SIGN, O.extract_number(grow_init)
  counts = ops.prune_regrow([dict(w=tw, momentum=tm, mask_bits=bits,
                                  dense_grad=tg, drop_noise=tn)], frac,
                            grow_init_mode=mode, grow_init_div=div,
                            initial_acc_scale=acc_scale)
  new_mask = ops.mask_unpack(bits, (n,)).cpu().numpy()
  return (new_mask, tw.cpu().numpy(), tm.cpu().numpy() if tm is not None else None,
          counts.cpu().numpy()[0])


def _diff(name, got, ref):
  got = np.asarray(got).reshape(-1)
  ref = np.asarray(ref).reshape(-1)
  bad = np.nonzero(got.view(np.uint32) != ref.view(np.uint32))[0]
  assert bad.size == 0, '%s: %d/%d differ, first at %s got %s ref %s' % (
      name, bad.size, got.size, bad[:8], got[bad[:8]], ref[bad[:8]])


def test_golden_reference_cases_one_by_one():
  from rigl_amd import ops
  z = np.load(os.path.join(G, 'update_cases.npz'))
  names = sorted({k.split('__')[0] for k in z.files if not k.startswith('generic_')})
  assert len(names) >= 25
  for n in names:
    g = lambda k: z['%s__%s' % (n, k)] if '%s__%s' % (n, k) in z.files else None
    nm, nw, nmom, counts = _run_rigl(ops, g('mask'), g('w'), g('g'), float(g('frac')),
                                     noise=g('noise'), mom=g('mom'),
                                     grow_init=str(g('grow_init')),
                                     acc_scale=float(g('acc_scale')))
    _diff(n + ' mask', nm, g('new_mask').astype(np.float32))
    _diff(n + ' weights', nw, g('new_w'))
    if g('mom') is not None:
      _diff(n + ' momentum', nmom, g('new_mom'))
    assert counts[0] == int(g('mask').sum())
    assert counts[4] == 0            # mask1 / mask2 disjoint
    assert counts[7] == counts[0]    # number of connections conserved


def _check_selections(ops, name, mask, w, g, frac, noise, mom, grow_init, acc_scale):
  """mask1, mask2 and the ordered top-k index lists of one update against the oracle's (sparse_optimizers_base.py:293-318)."""
  from rigl_amd import _lib
  n = mask.size
  tw, tg = _t(w), _t(g)
  tm = _t(mom) if mom is not None else None
  tn = _t(noise) if noise is not None else None
  bits = ops.mask_pack(_t(mask))
  mode, div = _lib.GROW_ZEROS, 1.0
  if grow_init.startswith('grad_scale'):
    mode, div = _lib.GROW_GRAD_SCALE, O.extract_number(grow_init)
  elif grow_init.startswith('grad_sign'):
    mode, div = _lib.GROW_GRAD_SIGN, O.extract_number(grow_init)
  out = ops.prune_regrow_selections(dict(w=tw, momentum=tm, mask_bits=bits, dense_grad=tg, drop_noise=tn), frac,
                                    grow_init_mode=mode, grow_init_div=div, initial_acc_scale=acc_scale)
  ref = O.rigl_mask_update(mask.reshape(-1), w.reshape(-1), g.reshape(-1), frac, noise=None if noise is None else noise.reshape(-1),
                           momentum=None if mom is None else mom.reshape(-1), grow_init=grow_init, initial_acc_scale=acc_scale)
  counts = out['counts'].cpu().numpy()
  assert counts[2] == ref['n_keep'] and counts[1] == ref['n_prune'], (name, counts, ref['n_keep'], ref['n_prune'])
  _diff(name + ' mask1', ops.mask_unpack(out['mask1_bits'], (n,)).cpu().numpy(), ref['mask1'])
  _diff(name + ' mask2', ops.mask_unpack(out['mask2_bits'], (n,)).cpu().numpy(), ref['mask2'])
  # the whole ordered lists (selected entries first, then the rest of the tensor in the same order) = the reference's top_k
  np.testing.assert_array_equal(out['idx1'].cpu().numpy(), np.asarray(ref['idx1'], dtype=np.int32), err_msg=name + ' idx1')
  np.testing.assert_array_equal(out['idx2'].cpu().numpy()[:ref['n_prune']], np.asarray(ref['idx2'], dtype=np.int32)[:ref['n_prune']],
                                err_msg=name + ' idx2 (selected part)')
  # ... and the update itself is the plain call's
  _diff(name + ' new mask', ops.mask_unpack(bits, (n,)).cpu().numpy(), ref['mask'].reshape(-1))
  _diff(name + ' new weights', tw.cpu().numpy(), ref['weights'].reshape(-1))
  return ref


def test_selections_and_topk_indices_on_the_golden_cases():
  """North star: "masks and top-k indices must be bit-exact" -- mask1 / mask2 and the ordered index lists of
  rigl_prune_regrow_selections on every reference-generated update case (missing #3)."""
  from rigl_amd import ops
  z = np.load(os.path.join(G, 'update_cases.npz'))
  names = sorted({k.split('__')[0] for k in z.files if not k.startswith('generic_')})
  for n in names:
    g = lambda k: z['%s__%s' % (n, k)] if '%s__%s' % (n, k) in z.files else None
    ref = _check_selections(ops, n, g('mask'), g('w'), g('g'), float(g('frac')), g('noise'), g('mom'), str(g('grow_init')),
                            float(g('acc_scale')))
    _diff(n + ' golden mask', ref['mask'].reshape(-1), g('new_mask').astype(np.float32))     # the oracle is the golden's twin


def test_selections_with_heavy_ties_and_a_large_layer():
  from rigl_amd import ops
  rs = np.random.RandomState(5)
  # heavy ties: weights and gradients drawn from a handful of values, so both thresholds cut through long runs of equals
  n = 70001
  mask = (rs.rand(n) < 0.3).astype(np.float32)
  w = rs.choice([-0.5, -0.25, 0.25, 0.5, 1.0], size=n).astype(np.float32)
  g = rs.choice([0.0, 0.125, -0.125, 0.75], size=n).astype(np.float32)
  _check_selections(ops, 'ties', mask, w, g, 0.3, None, None, 'zeros', 0.0)
  # a ResNet-50 3x3 layer (2.36 M weights) at ERK-0.8 density with noise
  n = 3 * 3 * 512 * 512
  mask = (rs.rand(n) < 0.2).astype(np.float32)
  w = (rs.randn(n) * 0.05).astype(np.float32)
  g = (rs.randn(n) * 1e-3).astype(np.float32)
  noise = (rs.randn(n) * 1e-4).astype(np.float32)
  _check_selections(ops, 'big', mask, w, g, 0.27, noise, (rs.randn(n) * 1e-2).astype(np.float32), 'zeros', 0.0)


def test_golden_reference_cases_batched_single_call():
  """All golden layers in ONE rigl_prune_regrow call (segmented launches)."""
  from rigl_amd import ops
  z = np.load(os.path.join(G, 'update_cases.npz'))
  names = sorted({k.split('__')[0] for k in z.files if not k.startswith('generic_')})
  # one call needs one (drop_fraction, grow_init, acc_scale): group by them
  groups = {}
  for n in names:
    key = (float(z[n + '__frac']), str(z[n + '__grow_init']), float(z[n + '__acc_scale']))
    groups.setdefault(key, []).append(n)
  for (frac, gi, acc), ns in groups.items():
    if gi != 'zeros':
      continue
    layers, keep = [], []
    for n in ns:
      has = lambda k: ('%s__%s' % (n, k)) in z.files
      l = dict(w=_t(z[n + '__w']), dense_grad=_t(z[n + '__g']),
               mask_bits=ops.mask_pack(_t(z[n + '__mask'])),
               momentum=_t(z[n + '__mom']) if has('mom') else None,
               drop_noise=_t(z[n + '__noise']) if has('noise') else None)
      layers.append(l)
      keep.append(n)
    ops.prune_regrow(layers, frac, initial_acc_scale=acc)
    for n, l in zip(keep, layers):
      size = z[n + '__mask'].size
      _diff(n + ' mask(batched)', ops.mask_unpack(l['mask_bits'], (size,)).cpu().numpy(),
            z[n + '__new_mask'].astype(np.float32))
      _diff(n + ' w(batched)', l['w'].cpu().numpy(), z[n + '__new_w'])
      if l['momentum'] is not None:
        _diff(n + ' mom(batched)', l['momentum'].cpu().numpy(), z[n + '__new_mom'])


def test_generic_scores_set_static():
  """Explicit score pointers: SET (uniform grow scores) and Static
  (score_grow = mask, reinit_when_same=True)."""
  from rigl_amd import _lib, ops
  z = np.load(os.path.join(G, 'update_cases.npz'))
  for n in ['generic_set_uniform', 'generic_static', 'generic_set_ties']:
    g = lambda k: z['%s__%s' % (n, k)]
    tw, tm = _t(g('w')), _t(g('mom'))
    bits = ops.mask_pack(_t(g('mask')))
    ops.prune_regrow([dict(w=tw, momentum=tm, mask_bits=bits, score_drop=_t(g('score_drop')),
                           score_grow=_t(g('score_grow')))], float(g('frac')),
                     momentum_reset_mode=_lib.MOMRESET_ZEROS,
                     reinit_when_same=bool(g('reinit')))
    _diff(n + ' mask', ops.mask_unpack(bits, (g('mask').size,)).cpu().numpy(), g('new_mask'))
    _diff(n + ' w', tw.cpu().numpy(), g('new_w'))
    _diff(n + ' mom', tm.cpu().numpy(), g('new_mom'))


@pytest.mark.parametrize('shape,sparsity,frac', [
    ((3, 3, 512, 512), 0.957, 0.3),      # largest ResNet-50 tensor (2 359 296)
    ((1, 1, 1024, 2048), 0.757, 0.3),
    ((2048, 1000), 0.852, 0.3),
    ((7, 7, 3, 64), 0.143, 0.3),
    ((3, 3, 512, 512), 0.998, 0.3),      # ERK 0.99 stress
    ((1, 1, 64, 64), 0.0, 0.3),          # dense layer in ERK 0.8
    ((784, 300), 0.9, 0.2999),
])
def test_resnet50_layer_sizes_vs_oracle(shape, sparsity, frac):
  from rigl_amd import ops
  rs = np.random.RandomState(hash(shape) % (2**31))
  n = int(np.prod(shape))
  w = (rs.randn(n) * 0.05).astype(np.float32)
  g = (rs.randn(n) * 1e-3).astype(np.float32)
  mom = rs.randn(n).astype(np.float32)
  noise = (rs.randn(n) * 1e-5).astype(np.float32)
  mask = O.get_mask_random_numpy((n,), sparsity, rs).astype(np.float32)
  w[rs.rand(n) < 0.05] = 0.0          # grown-but-untrained zeros: real ties
  ref = O.rigl_mask_update(mask, w, g, frac, noise=noise, momentum=mom)
  nm, nw, nmom, counts = _run_rigl(ops, mask, w, g, frac, noise=noise, mom=mom)
  _diff('mask', nm, ref['mask'])
  _diff('weights', nw, ref['weights'])
  _diff('momentum', nmom, ref['momentum'])
  assert (counts[0], counts[1], counts[2]) == (ref['n_ones'], ref['n_prune'], ref['n_keep'])


def test_heavy_ties_no_noise_large():
  """Quantised weights / grads, no noise: thousands of exact ties straddle both
  thresholds and many 4096-element chunks -> exercises the index-ordered
  tie ranks across workgroups."""
  from rigl_amd import ops
  rs = np.random.RandomState(7)
  n = 300007
  w = (rs.randint(-3, 4, size=n) / 8.0).astype(np.float32)
  g = (rs.randint(-2, 3, size=n) / 16.0).astype(np.float32)
  mask = O.get_mask_random_numpy((n,), 0.6, rs).astype(np.float32)
  for frac in (0.1, 0.5, 0.77):
    ref = O.rigl_mask_update(mask, w, g, frac)
    nm, nw, _, counts = _run_rigl(ops, mask, w, g, frac)
    _diff('mask f=%g' % frac, nm, ref['mask'])
    _diff('w f=%g' % frac, nw, ref['weights'])
    assert counts[5] > 0 and counts[6] > 0


def test_topk_mask():
  from rigl_amd import ops
  rs = np.random.RandomState(3)
  for n, k in [(1000, 0), (1000, 1000), (1000, 1), (5000, 1234), (100000, 99999), (33, 7)]:
    s = (rs.randint(0, 50, size=n) / 7.0).astype(np.float32)
    bits = ops.topk_mask(_t(s), k)
    got = ops.mask_unpack(bits, (n,)).cpu().numpy()
    ref = np.zeros(n, np.float32)
    ref[O.topk_order(s)[:k]] = 1
    _diff('topk n=%d k=%d' % (n, k), got, ref)


@pytest.mark.parametrize('ties', [2, 5, 130, 9000])
def test_topk_threshold_admits_all_some_or_one_of_its_ties(ties):
  """The tie-rank pass is skipped when the threshold key admits ALL of its duplicates (r == ties) and must
  run when only some are admitted (r < ties): plant `ties` copies of one value -- scattered over several
  4096-element chunks, some sharing a 4-element quad -- in otherwise distinct scores and cut at every
  interesting rank."""
  from rigl_amd import ops
  rs = np.random.RandomState(ties)
  n = 50021
  s = rs.permutation(n).astype(np.float32) + 1000.0           # distinct, all above / below the planted value
  pos = np.sort(rs.choice(n - 8, size=ties, replace=False))
  pos[:min(ties, 3)] = pos[0] - pos[0] % 4 + np.arange(min(ties, 3))   # a few ties inside one quad
  pos = np.unique(pos)
  ties = pos.size
  n_above = 20000
  order = rs.permutation(np.setdiff1d(np.arange(n), pos))
  s[order[:n_above]] += 1e6                                   # exactly n_above scores above the planted value
  s[order[n_above:]] -= 1e6 + 2000.0
  s[pos] = 777.0
  for k in (n_above, n_above + 1, n_above + ties - 1, n_above + ties, n_above + ties + 1, n_above + ties // 2):
    bits = ops.topk_mask(_t(s), k)
    got = ops.mask_unpack(bits, (n,)).cpu().numpy()
    ref = np.zeros(n, np.float32)
    ref[O.topk_order(s)[:k]] = 1
    _diff('ties=%d k=%d' % (ties, k), got, ref)
    admitted = int(got[pos].sum())
    assert admitted == min(max(k - n_above, 0), ties)
    if 0 < admitted < ties:
      np.testing.assert_array_equal(np.nonzero(got[pos])[0], np.arange(admitted))   # lowest indices win


def test_mask_pack_roundtrip_and_layout():
  from rigl_amd import ops
  rs = np.random.RandomState(0)
  for n in (1, 31, 32, 33, 63, 64, 65, 4096, 100003):
    m = (rs.rand(n) < 0.3).astype(np.float32)
    bits = ops.mask_pack(_t(m))
    words = bits.cpu().numpy().view(np.uint32)
    ref = np.packbits(m.astype(np.uint8), bitorder='little')
    ref = np.concatenate([ref, np.zeros((-len(ref)) % 4, np.uint8)]).view(np.uint32)
    np.testing.assert_array_equal(words, ref)      # bit i&31 of word i>>5, tail zero
    np.testing.assert_array_equal(ops.mask_unpack(bits, (n,)).cpu().numpy(), m)
