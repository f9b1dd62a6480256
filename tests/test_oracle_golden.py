"""Pins the CPU oracle (oracle/rigl_oracle.py) to golden vectors that were
produced by executing the reference's own code (tests/golden/make_golden.py)
and to the known answers of the reference's unit tests."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest

from oracle import rigl_oracle as O

G = os.path.join(os.path.dirname(__file__), 'golden')


def _unpack(hexbits, n):
  b = np.frombuffer(bytes.fromhex(hexbits), dtype=np.uint8)
  return np.unpackbits(b, bitorder='little')[:n]


# ---------------------------------------------------------------- mask init
def test_mask_random_matches_reference_bits():
  cases = json.load(open(os.path.join(G, 'mask_random.json')))
  for c in cases:
    n = int(np.prod(c['shape']))
    if c['seed'] == 'global11':
      np.random.seed(11)
      m = O.get_mask_random_numpy(c['shape'], c['sparsity'])
    else:
      m = O.get_mask_random_numpy(c['shape'], c['sparsity'],
                                  np.random.RandomState(c['seed']))
    assert m.shape == tuple(c['shape'])
    assert int(m.sum()) == c['ones']
    np.testing.assert_array_equal(m.reshape(-1).astype(np.uint8),
                                  _unpack(c['bits'], n))


@pytest.mark.parametrize('shape,sparsity,expected_ones',
                         [((30, 4), 0.5, 60), ((1, 2, 1, 4), 0.8, 2),
                          ((30,), 0.1, 27)])
def test_mask_fraction_known_answers(shape, sparsity, expected_ones):
  # rigl/sparse_utils_test.py:47-55
  assert O.get_mask_random_numpy(shape, sparsity).sum() == expected_ones


@pytest.mark.parametrize('shape,sparsity', [((30, 40), 0.5),
                                            ((1, 2, 1, 4), 0.8), ((3,), 0.1)])
def test_mask_connection_determinism(shape, sparsity):
  # rigl/sparse_utils_test.py:37-45
  a = O.get_mask_random_numpy(shape, sparsity)
  b = O.get_mask_random_numpy(shape, sparsity)
  assert a.sum() == b.sum()


# ---------------------------------------------------------------- sparsities
def test_sparsities_match_reference():
  data = json.load(open(os.path.join(G, 'sparsities.json')))
  for r in data['runs']:
    masks = OrderedDict((n, tuple(s)) for n, s in zip(r['names'], r['shapes']))
    got = O.get_sparsities(masks, r['method'], r['default_sparsity'],
                           r['custom'], erk_power_scale=r['erk_power_scale'])
    for name, ref_hex in zip(r['names'], r['sparsities']):
      assert float(got[name]) == float.fromhex(ref_hex), (r['net'], name)


def test_sparsities_errors_match_reference():
  data = json.load(open(os.path.join(G, 'sparsities.json')))
  masks = OrderedDict([('var1/mask:0', (2, 4)), ('var2/mask:0', (2, 3)),
                       ('var3/mask:0', (1, 1, 3))])
  for e in data['errors']:
    with pytest.raises(ValueError) as ei:
      O.get_sparsities(masks, e['method'], 0.5, e['custom'])
    assert str(ei.value) == e['error']


@pytest.mark.parametrize('shape1,shape2,s',
                         [((2, 3), (2, 3), 0.5), ((1, 1, 2, 3), (1, 1, 2, 3), 0.3),
                          ((8, 6), (4, 3), 0.7), ((80, 4), (20, 20), 0.8),
                          ((2, 6), (2, 3), 0.8)])
def test_erdos_renyi_scale(shape1, shape2, s):
  # rigl/sparse_utils_test.py:108-143
  masks = OrderedDict([('var1/mask:0', shape1), ('var2/mask:0', shape2)])
  sp = O.get_sparsities(masks, 'erdos_renyi', s, {})
  n1, n2 = np.prod(shape1), np.prod(shape2)
  uni = O.get_n_zeros(n1, s) + O.get_n_zeros(n2, s)
  cur = (O.get_n_zeros(n1, sp['var1/mask:0']) +
         O.get_n_zeros(n2, sp['var2/mask:0']))
  assert abs(uni - cur) <= 2
  f1 = (shape1[-1] + shape1[-2]) / float(shape1[-1] * shape1[-2])
  f2 = (shape2[-1] + shape2[-2]) / float(shape2[-1] * shape2[-2])
  assert abs((1 - sp['var1/mask:0']) / f1 - (1 - sp['var2/mask:0']) / f2) < 1e-7


def test_resnet50_erk_matches_published_ratios():
  # README.md:65 (0.42x FLOPs at ERK 0.8); derived totals BASELINE.md section 2
  data = json.load(open(os.path.join(G, 'sparsities.json')))
  r = data['runs'][0]
  assert (r['net'], r['method'], r['default_sparsity']) == (
      'resnet50', 'erdos_renyi_kernel', 0.8)
  n_tot = nnz = 0
  for sh, hx in zip(r['shapes'], r['sparsities']):
    n = int(np.prod(sh))
    n_tot += n
    nnz += n - O.get_n_zeros(n, float.fromhex(hx))
  assert n_tot == 25502912
  assert nnz == 5100630


# ---------------------------------------------------------------- schedule
def test_rigl_global_step_sequences():
  # rigl/sparse_optimizers_test.py:349-367
  sched = json.load(open(os.path.join(G, 'schedule.json')))
  assert len(sched['rigl_increment']) == 3
  for c in sched['rigl_increment']:
    s = O.RigLSchedule(c['begin'], c['end'], c['freq'], 0.5)
    seq = []
    for _ in c['seq']:
      before = s.global_step
      s.step()
      seq.append(s.global_step - before)
    assert seq == c['seq']


def test_set_update_iterations():
  sched = json.load(open(os.path.join(G, 'schedule.json')))
  for c in sched['set_updates']:
    s = O.SETSchedule(c['begin'], c['end'], c['freq'], 0.5)
    changed = [i for i in range(1, c['iters'] + 1) if s.step()[0]]
    assert changed == c['changed']


def test_drop_fraction_bits():
  sched = json.load(open(os.path.join(G, 'schedule.json')))
  for c in sched['drop_fraction']:
    for step, flag, hx in c['values']:
      got = O.get_drop_fraction(c['anneal'], c['init'], step, c['begin'],
                                c['end'], flag)
      assert np.float32(got) == np.float32(float.fromhex(hx)), (c, step)
  with pytest.raises(ValueError) as ei:
    O.get_drop_fraction('bogus', 0.3, 0, 0, 5, True)
  assert str(ei.value) == sched['bad_anneal_error']


# ---------------------------------------------------------------- update core
def _cases(prefix_filter):
  z = np.load(os.path.join(G, 'update_cases.npz'))
  names = sorted({k.split('__')[0] for k in z.files})
  return z, [n for n in names if prefix_filter(n)]


def test_rigl_update_matches_reference_bits():
  z, names = _cases(lambda n: not n.startswith('generic_'))
  assert len(names) >= 25
  for n in names:
    g = lambda k: z['%s__%s' % (n, k)] if '%s__%s' % (n, k) in z.files else None
    r = O.rigl_mask_update(g('mask'), g('w'), g('g'), g('frac'),
                           noise=g('noise'), momentum=g('mom'),
                           grow_init=str(g('grow_init')),
                           initial_acc_scale=float(g('acc_scale')))
    np.testing.assert_array_equal(r['mask'], g('new_mask'), err_msg=n)
    np.testing.assert_array_equal(r['weights'].view(np.uint32),
                                  g('new_w').view(np.uint32), err_msg=n)
    if g('mom') is not None:
      np.testing.assert_array_equal(r['momentum'].view(np.uint32),
                                    g('new_mom').view(np.uint32), err_msg=n)
    # invariants of the reference tests
    assert r['mask'].sum() == g('mask').sum()             # :94-118
    # NB: an inactive entry can also enter through mask1 when its noise
    # outranks a tiny active weight (reference behaviour, Appendix A) -- only
    # mask2-grown connections are re-initialised.
    grown = np.logical_and(g('mask').reshape(-1) == 0, r['mask2'] == 1)
    if str(g('grow_init')) == 'zeros':
      assert np.all(r['weights'].reshape(-1)[grown] == 0)  # :141-156


def test_generic_update_matches_reference_bits():
  z, names = _cases(lambda n: n.startswith('generic_'))
  assert len(names) == 3
  for n in names:
    g = lambda k: z['%s__%s' % (n, k)]
    r = O.get_update(g('score_drop'), g('score_grow'), g('mask'), g('w'),
                     g('frac'), momentum=g('mom'),
                     reinit_when_same=bool(g('reinit')),
                     rigl_momentum_reset=False)
    np.testing.assert_array_equal(r['mask'], g('new_mask'), err_msg=n)
    np.testing.assert_array_equal(r['weights'], g('new_w'), err_msg=n)
    np.testing.assert_array_equal(r['momentum'], g('new_mom'), err_msg=n)


def test_static_mask_never_changes():
  # rigl/sparse_optimizers_test.py:225-244 (score_grow = mask, reinit)
  z, _ = _cases(lambda n: True)
  np.testing.assert_array_equal(z['generic_static__mask'],
                                z['generic_static__new_mask'])


def test_trajectory_matches_reference():
  """Full RigL training trajectory of the reference's toy FC problem
  (sparse_optimizers_test.py:299-328) through minimize()."""
  t = np.load(os.path.join(G, 'trajectory.npz'))
  for tag, inner, acc in [('rigl_mom', 'mom', 0.0), ('rigl_mom_acc', 'mom', 0.5),
                          ('rigl_sgd', 'sgd', 0.0)]:
    W, M, GS, FR = (t[tag + '__w'], t[tag + '__mask'], t[tag + '__gs'],
                    t[tag + '__frac'])
    n_inp, n_out = W.shape[1:]
    w, m = W[0].copy(), M[0].copy()
    a = np.zeros_like(w)
    sched = O.RigLSchedule(1, 17, 4, 0.4, 'cosine')
    for i in range(len(FR)):
      gs = sched.global_step
      dense = np.broadcast_to(
          (np.arange(n_out, dtype=np.float32) * np.float32(gs)).astype(
              np.float32), (n_inp, n_out)).astype(np.float32)
      is_upd, frac = sched.step()
      assert np.float32(frac) == FR[i]
      if is_upd:
        r = O.rigl_mask_update(m, w, dense, frac, noise=None,
                               momentum=a if inner == 'mom' else None,
                               initial_acc_scale=acc)
        m, w = r['mask'], r['weights']
        if inner == 'mom':
          a = r['momentum']
      else:
        g = O.masked_grad(dense, m, w, 0.0)
        if inner == 'mom':
          w, a = O.momentum_apply(w, a, g, 0.01, 0.9, nesterov=True)
        else:
          w = O.sgd_apply(w, g, 0.01)
      assert sched.global_step == GS[i + 1]
      np.testing.assert_array_equal(m, M[i + 1], err_msg='%s step %d' % (tag, i))
      np.testing.assert_array_equal(w.view(np.uint32), W[i + 1].view(np.uint32),
                                    err_msg='%s step %d' % (tag, i))
      if inner == 'mom':
        np.testing.assert_array_equal(a, t[tag + '__mom'][i])


# ---------------------------------------------------------------- misc
@pytest.mark.parametrize('method', ['ones', 'zero', None, 0])
def test_grow_tensor_value_error(method):
  # rigl/sparse_optimizers_test.py:181-189
  with pytest.raises(ValueError):
    O.get_grow_tensor_rigl(np.zeros((3, 4), np.float32),
                           np.zeros((3, 4), np.float32), method)


def test_extract_number():
  # rigl/sparse_optimizers_base.py:45-59 docstring examples
  assert O.extract_number('foo_.5') == 0.5
  assert O.extract_number('foo_foo.5') == 1.0
  assert O.extract_number('foo_0.5') == 0.5
  assert O.extract_number('foo_4') == 4
  assert O.extract_number('zeros') == 1.0


def test_topk_order_ties_lower_index_first():
  x = np.array([1., 3., 3., 0., -0., 3., 1.], np.float32)
  assert O.topk_order(x).tolist() == [1, 2, 5, 0, 6, 3, 4]


def test_c_oracle_agrees_with_numpy_oracle_and_reference():
  """oracle/c (qsort, full-sort form) vs oracle/rigl_oracle.py vs the
  reference-generated goldens."""
  from oracle import c_oracle
  z, names = _cases(lambda n: not n.startswith('generic_'))
  for n in names:
    g = lambda k: z['%s__%s' % (n, k)] if '%s__%s' % (n, k) in z.files else None
    if str(g('grow_init')) != 'zeros':
      continue
    sd, sg = O.rigl_scores(g('mask'), g('w'), g('g'), g('noise'))
    mv = None if g('mom') is None else (g('g') * np.float32(g('acc_scale'))).astype(np.float32)
    nm, nw, nmom, counts = c_oracle.update(sd, sg, g('mask'), g('w'), float(g('frac')), momentum=g('mom'),
                                           momentum_values=mv)
    np.testing.assert_array_equal(nm, g('new_mask').reshape(-1), err_msg=n)
    np.testing.assert_array_equal(nw.view(np.uint32), g('new_w').reshape(-1).view(np.uint32), err_msg=n)
    if g('mom') is not None:
      np.testing.assert_array_equal(nmom, g('new_mom').reshape(-1), err_msg=n)
    assert counts[4] == 0


def test_resnet_cpu_restatement_float64_twin_and_update_rule():
  """oracle/resnet_cpu.py: the float64 evaluation (what tests/test_k1_fp32_gpu.py holds the fp32 kernels against) is
  the same model as the float32 one, and one train_step applies TF's Nesterov ApplyMomentum to EVERY trainable
  variable -- masked kernels with mask * dense + wd * W, batch-norm parameters and the fc bias without a regulariser
  (imagenet_train_eval.py:360-361; O.momentum_apply is the bit-level statement of the same rule)."""
  torch = pytest.importorskip('torch')
  from oracle.resnet_cpu import ResNet50CPU
  torch.set_num_threads(min(torch.get_num_threads(), 8))
  m32 = ResNet50CPU(sparsity_by_layer=[0.8] * 54, seed=3, num_classes=10)
  m64 = ResNet50CPU(sparsity_by_layer=[0.8] * 54, seed=3, num_classes=10, dtype=torch.float64)
  with torch.no_grad():
    for a, b in zip(m32.w, m64.w):
      b.copy_(a.double())
  for m in (m32, m64):
    assert all(t.dtype == m.dtype for t in m.w + m.m + m.other_params())
  for a, b in zip(m32.m, m64.m):
    np.testing.assert_array_equal(a.numpy(), b.numpy())
  x = torch.randn(2, 3, 64, 64)
  y = torch.tensor([1, 7])
  w0 = [w.detach().clone() for w in m32.w]
  o0 = [p.detach().clone() for p in m32.other_params()]
  l32 = m32.train_step(x, y, lr=0.1, mu=0.9, wd=1e-4, keep_dense=True)
  l64 = m64.train_step(x.double(), y, lr=0.1, mu=0.9, wd=1e-4, keep_dense=True)
  assert abs(l32 - l64) <= 1e-5 * abs(l64)
  # the update, restated: accum starts at zero, so accum' = g and W' = W - lr * g - lr * mu * g
  for i in (0, 5, 53):
    g = m32.m[i].numpy() * m32.dense_grads[i] + np.float32(1e-4) * w0[i].numpy()
    w_ref, a_ref = O.momentum_apply(w0[i].numpy(), np.zeros_like(g), g, 0.1, 0.9)
    np.testing.assert_allclose(m32.w[i].detach().numpy(), w_ref, rtol=0, atol=2e-7 * np.abs(w_ref).max())
    np.testing.assert_allclose(m32.mom[i].numpy(), a_ref, rtol=0, atol=1e-6 * np.abs(a_ref).max())
  moved = [float((p.detach() - q).abs().max()) for p, q in zip(m32.other_params(), o0)]
  # the fc bias and the stem's batch norm are trained (inside the residual branches the zero-initialised last gamma
  # of every block leaves the branch's own parameters without a gradient at step 0)
  assert moved[-1] > 0 and moved[0] > 0 and moved[1] > 0
  assert float(m32.mom_other[-1].abs().max()) > 0
