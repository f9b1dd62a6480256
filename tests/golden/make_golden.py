#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by EXECUTING THE REFERENCE.

Runs only where /root/reference exists (the build container).  It imports the
reference's own ``rigl/sparse_utils.py`` and ``rigl/sparse_optimizers_base.py``
unmodified, with TensorFlow replaced by the NumPy shim in ``tf_shim.py``
(TensorFlow itself is not installable here), drives them on seeded inputs and
stores inputs + outputs.  The committed outputs are what the oracle
(``oracle/rigl_oracle.py``) and, through it, the HIP kernels are pinned to.

  python tests/golden/make_golden.py        # rewrites the fixtures
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('RIGL_REFERENCE', '/root/reference')
sys.path.insert(0, HERE)

import tf_shim  # noqa: E402
import layer_shapes  # noqa: E402

tf_shim.install()
# Make ``import rigl.*`` resolve to the reference tree (not to the alias
# package of this repo).
_pkg = types.ModuleType('rigl')
_pkg.__path__ = [os.path.join(REF, 'rigl')]
sys.modules['rigl'] = _pkg

from rigl import sparse_utils as ref_su  # noqa: E402
from rigl import sparse_optimizers_base as ref_base  # noqa: E402
from rigl import str_sparsities as ref_str  # noqa: E402
from tensorflow.python.training import optimizer as shim_opt  # noqa: E402

V = tf_shim.Variable


def fhex(x):
  return float(x).hex()


# ----------------------------------------------------------------------------
# A. get_mask_random_numpy
# ----------------------------------------------------------------------------
def gen_mask_random():
  cases = []
  specs = [((30, 4), 0.5, 0), ((1, 2, 1, 4), 0.8, 1), ((30,), 0.1, 2),
           ((3, 3, 64, 64), 0.8, 0), ((784, 300), 0.9, 7),
           ((1, 1, 64, 256), 0.0, 3), ((5, 7), 1.0, 4), ((1, 1, 256, 64), 0.513, 5)]
  for shape, sparsity, seed in specs:
    rs = np.random.RandomState(seed)
    m = ref_su.get_mask_random_numpy(list(shape), sparsity, random_state=rs)
    cases.append(dict(shape=list(shape), sparsity=sparsity, seed=seed,
                      ones=int(m.sum()),
                      bits=np.packbits(m.astype(np.uint8).reshape(-1),
                                       bitorder='little').tobytes().hex()))
  # global-RNG path + get_mask_random wrapper (tf.constant of dtype)
  np.random.seed(11)
  mk = V(np.ones((6, 5), np.float32), 'a/mask')
  m = ref_su.get_mask_random(mk, 0.4, np.int32)
  cases.append(dict(shape=[6, 5], sparsity=0.4, seed='global11',
                    ones=int(m.sum()), dtype=str(m.dtype),
                    bits=np.packbits(m.astype(np.uint8).reshape(-1),
                                     bitorder='little').tobytes().hex()))
  return cases


# ----------------------------------------------------------------------------
# B. sparsity distributions
# ----------------------------------------------------------------------------
def _masks(shapes):
  return [V(np.ones(s, np.float32), n) for n, s in shapes.items()]


def gen_sparsities():
  out = []
  nets = dict(
      resnet50=layer_shapes.resnet50(),
      resnet50_dense_stem=layer_shapes.resnet50(prune_first_layer=False),
      mobilenet_v1=layer_shapes.mobilenet_v1(),
      wrn_22_1=layer_shapes.wide_resnet(22, 1),
      wrn_16_4=layer_shapes.wide_resnet(16, 4),
      mnist=layer_shapes.mnist_mlp(),
      t1=dict([('var1/mask:0', (2, 4)), ('var2/mask:0', (2, 3)),
               ('var3/mask:0', (1, 1, 3))]),
      t2=dict([('var1/mask:0', (80, 4)), ('var2/mask:0', (20, 20))]),
      t3=dict([('var1/mask:0', (8, 6)), ('var2/mask:0', (4, 3))]),
  )
  runs = [
      ('resnet50', 'erdos_renyi_kernel', 0.8, {}, 1.0),
      ('resnet50', 'erdos_renyi_kernel', 0.9, {}, 1.0),
      ('resnet50', 'erdos_renyi', 0.8, {}, 1.0),
      ('resnet50', 'random', 0.8, {}, 1.0),
      ('resnet50', 'erdos_renyi_kernel', 0.8, {}, 0.5),
      ('resnet50_dense_stem', 'erdos_renyi_kernel', 0.99, {}, 1.0),
      ('resnet50', 'erdos_renyi_kernel', 0.95, {'initial_conv': 0.0,
                                                'final_dense': 0.8}, 1.0),
      ('mobilenet_v1', 'random', 0.9, {}, 1.0),
      ('mobilenet_v1', 'erdos_renyi_kernel', 0.9, {}, 1.0),
      ('wrn_22_1', 'erdos_renyi_kernel', 0.8, {}, 1.0),
      ('wrn_16_4', 'random', 0.9, {}, 1.0),
      ('wrn_16_4', 'erdos_renyi', 0.9, {}, 1.0),
      ('mnist', 'random', 0.9, {'layer3': 0.0}, 1.0),
      ('mnist', 'erdos_renyi', 0.9, {}, 1.0),
      ('t1', 'erdos_renyi', 0.4, {'var3': 0.8}, 1.0),
      ('t1', 'random', 0.4, {'var1': 0.8}, 1.0),
      ('t2', 'erdos_renyi', 0.8, {}, 1.0),
      ('t3', 'erdos_renyi', 0.7, {}, 1.0),
  ]
  for net, method, s, custom, power in runs:
    custom_full = {('resnet_model/' + k if net.startswith(('resnet', 'mobile',
                                                            'wrn')) else k): v
                   for k, v in custom.items()}
    masks = _masks(nets[net])
    res = ref_su.get_sparsities(masks, method, s, custom_full,
                                erk_power_scale=power)
    out.append(dict(net=net, method=method, default_sparsity=s,
                    custom=custom_full, erk_power_scale=power,
                    names=[m.name for m in masks],
                    shapes=[list(m.shape) for m in masks],
                    sparsities=[fhex(res[m.name]) for m in masks]))
  # error cases
  errs = []
  for method, custom in [('bogus', {}), ('random', {'nope': 0.5})]:
    try:
      ref_su.get_sparsities(_masks(nets['t1']), method, 0.5, custom)
      errs.append(dict(method=method, custom=custom, error=None))
    except ValueError as e:
      errs.append(dict(method=method, custom=custom, error=str(e)))
  # The reference's own per-layer ResNet-50 size table (str_sparsities.py).
  table = []
  for l in ref_str.REPORTED_SPARSITIES.strip().split('\n'):
    f = l.split('-')[1].strip().split(' ')
    if f[0] != 'Overall':
      table.append([ref_str._name_map_str(f[0]), int(f[1]), int(f[2])])
  return dict(runs=out, errors=errs, resnet50_size_table=table)


# ----------------------------------------------------------------------------
# C. the prune/regrow core, through the reference's own classes
# ----------------------------------------------------------------------------
class _Getters:
  """Stand-in for PruningGetterTf1Mixin (rigl/sparse_optimizers.py:46-56)."""

  def get_weights(self):
    return self._ws

  def get_masks(self):
    return self._ms

  def get_masked_weights(self):
    return [m.value * w.value for m, w in zip(self._ms, self._ws)]


class RefRigL(_Getters, ref_base.SparseRigLOptimizerBase):
  pass


class RefSET(_Getters, ref_base.SparseSETOptimizerBase):
  pass


def _noise_inject(z):
  tf_shim.NOISE_FOR_SEED.clear()
  if z is not None:
    tf_shim.NOISE_FOR_SEED[None] = lambda key, shape: np.asarray(
        z, np.float32).reshape(shape)


def run_ref_rigl_update(mask, w, g, frac, z=None, noise_std=1e-5, mom=None,
                        grow_init='zeros', acc_scale=0.):
  """Calls the reference's generic_mask_update on one layer."""
  tf_shim.STORE.reset()
  wv = V(np.array(w, np.float32), 'layer/weights')
  mv = V(np.array(mask, np.float32), 'layer/mask')
  inner = shim_opt.MomentumOptimizer(0.1, 0.9, lambda vl: [], True)
  inner._create_slots([wv])
  if mom is not None:
    inner.get_slot(wv, 'momentum').value = np.array(mom, np.float32)
  opt = RefRigL(inner, 0, 100, 1, drop_fraction=0.1, grow_init=grow_init,
                initial_acc_scale=acc_scale)
  opt._ws, opt._ms = [wv], [mv]
  opt._weight2masked_grads = {wv.name: np.array(g, np.float32)}
  opt.drop_fraction = np.float32(frac)
  opt._global_step = V(np.int64(0), 'global_step', dtype=np.int64)
  _noise_inject(z)
  opt.generic_mask_update(mv, wv, noise_std=noise_std)
  noise = None
  if z is not None:
    noise = (np.asarray(z, np.float32) * np.float32(noise_std)).astype(
        np.float32)
  return dict(new_mask=mv.value.copy(), new_w=wv.value.copy(),
              new_mom=inner.get_slot(wv, 'momentum').value.copy(),
              noise=noise)


def run_ref_generic_update(score_drop, score_grow, mask, w, frac,
                           reinit_when_same=False, mom=None):
  """Calls the reference's SET-base _get_update_op with explicit scores."""
  tf_shim.STORE.reset()
  wv = V(np.array(w, np.float32), 'layer/weights')
  mv = V(np.array(mask, np.float32), 'layer/mask')
  inner = shim_opt.MomentumOptimizer(0.1, 0.9, lambda vl: [], True)
  inner._create_slots([wv])
  if mom is not None:
    inner.get_slot(wv, 'momentum').value = np.array(mom, np.float32)
  opt = RefSET(inner, 0, 100, 1, drop_fraction=0.1)
  opt._ws, opt._ms = [wv], [mv]
  opt.drop_fraction = np.float32(frac)
  opt._global_step = V(np.int64(0), 'global_step', dtype=np.int64)
  opt._get_update_op(np.array(score_drop, np.float32),
                     np.array(score_grow, np.float32), mv, wv,
                     reinit_when_same=reinit_when_same)
  return dict(new_mask=mv.value.copy(), new_w=wv.value.copy(),
              new_mom=inner.get_slot(wv, 'momentum').value.copy())


def gen_update_cases():
  rs = np.random.RandomState(1234)
  cases = {}

  def add(name, shape, frac, sparsity=0.5, wgen=None, ggen=None, noise=True,
          mom=True, grow_init='zeros', acc_scale=0., mask=None,
          noise_std=1e-5):
    n = int(np.prod(shape))
    w = (wgen(n) if wgen else rs.randn(n)).astype(np.float32).reshape(shape)
    g = (ggen(n) if ggen else rs.randn(n)).astype(np.float32).reshape(shape)
    if mask is None:
      mask = ref_su.get_mask_random_numpy(list(shape), sparsity,
                                          random_state=rs)
    mask = np.asarray(mask, np.float32).reshape(shape)
    z = rs.randn(n).astype(np.float32).reshape(shape) if noise else None
    m0 = rs.randn(n).astype(np.float32).reshape(shape) if mom else None
    r = run_ref_rigl_update(mask, w, g, frac, z=z, mom=m0,
                            grow_init=grow_init, acc_scale=acc_scale,
                            noise_std=noise_std)
    c = dict(mask=mask, w=w, g=g, frac=np.float32(frac),
             new_mask=r['new_mask'], new_w=r['new_w'],
             grow_init=np.array(grow_init), acc_scale=np.float32(acc_scale))
    if z is not None:
      c['noise'] = r['noise']
    if m0 is not None:
      c['mom'] = m0
      c['new_mom'] = r['new_mom']
    for k, v in c.items():
      cases['%s__%s' % (name, k)] = v

  q = lambda lv: (lambda n: rs.randint(-lv, lv + 1, size=n) / 4.0)
  add('rand1k', (1000,), 0.3)
  add('rand_fc', (30, 40), 0.5, sparsity=0.9)
  add('conv3x3', (3, 3, 16, 32), 0.3, sparsity=0.8)
  add('conv_big', (3, 3, 64, 64), 0.3, sparsity=0.8)
  add('odd_len', (4099,), 0.3, sparsity=0.77)
  add('ties_w', (4096,), 0.3, wgen=q(3), noise=False)
  add('ties_g', (4096,), 0.3, ggen=q(2))
  add('ties_both', (5000,), 0.45, wgen=q(2), ggen=q(2), noise=False)
  add('zero_grad', (2048,), 0.3, ggen=lambda n: np.zeros(n))
  add('zero_w', (2048,), 0.3, wgen=lambda n: np.zeros(n), noise=False)
  add('zero_w_noise', (2048,), 0.3, wgen=lambda n: np.zeros(n))
  add('frac0', (777,), 0.0)
  add('frac1', (777,), 1.0)
  add('frac_small', (777,), 0.001)
  add('all_ones', (640,), 0.3, mask=np.ones(640))
  add('all_zeros', (640,), 0.3, mask=np.zeros(640))
  add('one_active', (640,), 0.9, mask=np.eye(1, 640, 77).reshape(-1))
  add('tiny_w', (3000,), 0.3, wgen=lambda n: rs.randn(n) * 1e-6)
  add('neg_zero', (1024,), 0.3,
      wgen=lambda n: np.where(rs.rand(n) < 0.5, -0.0, rs.randn(n)),
      ggen=lambda n: np.where(rs.rand(n) < 0.5, -0.0, rs.randn(n)),
      noise=False)
  add('acc_scale', (1500,), 0.3, acc_scale=0.5)
  add('grad_scale', (1500,), 0.3, grow_init='grad_scale_2')
  add('grad_sign', (1500,), 0.3, grow_init='grad_sign_4')
  add('big_noise', (1500,), 0.3, noise_std=0.5)
  add('high_sparse', (20000,), 0.3, sparsity=0.99)
  add('size1', (1,), 0.5, mask=np.ones(1))
  add('size33', (33,), 0.5)

  # generic scores (SET: uniform grow scores; Static: score_grow = mask and
  # reinit_when_same=True -- rigl/sparse_optimizers.py:109-123)
  for name, n, frac, reinit in [('set_uniform', 3000, 0.3, False),
                                ('static', 3000, 0.3, True),
                                ('set_ties', 2000, 0.5, False)]:
    w = rs.randn(n).astype(np.float32)
    mask = ref_su.get_mask_random_numpy([n], 0.6, random_state=rs).astype(
        np.float32)
    sd = np.abs(mask * w) + (rs.randn(n) * 1e-5).astype(np.float32)
    if name == 'static':
      sg = mask.copy()
    elif name == 'set_ties':
      sg = (rs.randint(0, 4, size=n) / 4.0).astype(np.float32)
      sd = np.abs(mask * np.round(w * 2) / 2).astype(np.float32)
    else:
      sg = rs.rand(n).astype(np.float32)
    m0 = rs.randn(n).astype(np.float32)
    r = run_ref_generic_update(sd, sg, mask, w, frac, reinit_when_same=reinit,
                               mom=m0)
    for k, v in dict(mask=mask, w=w, score_drop=sd.astype(np.float32),
                     score_grow=sg, frac=np.float32(frac), mom=m0,
                     reinit=np.int32(reinit), new_mask=r['new_mask'],
                     new_w=r['new_w'], new_mom=r['new_mom']).items():
      cases['generic_%s__%s' % (name, k)] = v
  return cases


# ----------------------------------------------------------------------------
# D. schedule + drop fraction, E. a short trajectory through minimize()
# ----------------------------------------------------------------------------
def _fc_problem(n_inp, n_out, method, begin, end, freq, frac, anneal='constant',
                inner='sgd', lr=1e-3, seed=0, acc_scale=0., grow_init='zeros'):
  """The reference test's toy problem (sparse_optimizers_test.py:299-328):
  y = x @ (mask*W) with x = ones, loss = sum(y * arange(n_out) * global_step).
  Gradients are supplied analytically (the shim has no autodiff)."""
  tf_shim.STORE.reset()
  rs = np.random.RandomState(seed)
  wv = V(rs.randn(n_inp, n_out).astype(np.float32), 'fc/weights')
  mv = V((rs.rand(n_inp, n_out) < 0.5).astype(np.float32), 'fc/mask')
  gs = tf_shim.sys.modules[
      'tensorflow.python.training.training_util'].get_or_create_global_step()

  def dense_grad():
    scale = (np.arange(n_out, dtype=np.float32) *
             np.float32(gs.value)).astype(np.float32)
    return np.broadcast_to(scale, (n_inp, n_out)).astype(np.float32)

  def grad_fn(var_list):
    if var_list is None:      # gradient w.r.t. the raw variable: mask * dense
      return [((mv.value * dense_grad()).astype(np.float32), wv)]
    return [(dense_grad(), var_list[0])]   # w.r.t. mask*W: dense

  if inner == 'sgd':
    io = shim_opt.GradientDescentOptimizer(lr, grad_fn)
  else:
    io = shim_opt.MomentumOptimizer(lr, 0.9, grad_fn, use_nesterov=True)
  cls = RefRigL if method == 'rigl' else RefSET
  kw = dict(drop_fraction=frac, drop_fraction_anneal=anneal,
            grow_init=grow_init)
  if method == 'rigl':
    kw['initial_acc_scale'] = acc_scale
  opt = cls(io, begin, end, freq, **kw)
  opt._ws, opt._ms = [wv], [mv]
  return opt, io, wv, mv, gs


def gen_schedule():
  out = dict(rigl_increment=[], set_updates=[], drop_fraction=[])
  # RigL: golden 0/1 global-step increments (sparse_optimizers_test.py:349-367)
  expected = {
      (3, 7, 2): [1, 1, 1, 0, 1, 1, 0, 1, 1, 0, 1],
      (1, 5, 3): [1, 0, 1, 1, 1, 0, 1, 1, 1, 1, 1],
      (0, 4, 1): [0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 1],
  }
  for (b, e, f), exp in expected.items():
    opt, _, _, _, gs = _fc_problem(3, 5, 'rigl', b, e, f, 0.5)
    _noise_inject(None)
    seq = []
    for _ in exp:
      before = int(gs.value)
      tf_shim.run_op(opt.minimize(None, gs))
      seq.append(int(gs.value) - before)
    assert seq == exp, ('shim/reference disagree with the reference test',
                        (b, e, f), seq, exp)
    out['rigl_increment'].append(dict(begin=b, end=e, freq=f, seq=seq))
  # SET: which iterations change the mask (sparse_optimizers_test.py:71-118)
  for (b, e, f, iters) in [(1, 4, 2, 6), (0, 10, 3, 14), (2, -1, 4, 20)]:
    opt, _, _, mv, gs = _fc_problem(15, 25, 'set', b, e, f, 0.5)
    rs = np.random.RandomState(5)
    tf_shim.NOISE_FOR_SEED.clear()
    tf_shim.UNIFORM_FOR_SEED.clear()
    tf_shim.UNIFORM_FOR_SEED[None] = lambda key, shape: rs.rand(*shape)
    changed = []
    for i in range(1, iters + 1):
      m0 = mv.value.copy()
      tf_shim.run_op(opt.minimize(None, gs))
      assert m0.sum() == mv.value.sum()
      if not np.array_equal(m0, mv.value):
        changed.append(i)
    out['set_updates'].append(dict(begin=b, end=e, freq=f, iters=iters,
                                   changed=changed))
  tf_shim.UNIFORM_FOR_SEED.clear()
  # drop fraction values
  for anneal, b, e in [('constant', 0, 100), ('cosine', 0, 25000),
                       ('cosine', 1000, 25000), ('exponential_3', 0, 1000),
                       ('exponential_1', 10, 110), ('cosine', 0, 7)]:
    opt, _, _, _, gs = _fc_problem(3, 5, 'rigl', b, e, 1, 0.3, anneal=anneal)
    opt._begin_step = np.int64(b)
    opt._end_step = np.int64(e)
    vals = []
    steps = sorted(set([b, b + 1, (b + e) // 2, e - 1, e, e + 5, b + 100,
                        b + 12345 % max(e - b, 1)]))
    for s in steps:
      if s < 0:
        continue
      for flag in (True, False):
        v = opt.get_drop_fraction(np.int64(s), np.bool_(flag))
        vals.append([int(s), bool(flag), fhex(np.float32(v))])
    out['drop_fraction'].append(dict(anneal=anneal, begin=b, end=e, init=0.3,
                                     values=vals))
  try:
    opt, _, _, _, gs = _fc_problem(3, 5, 'rigl', 0, 5, 1, 0.3, anneal='bogus')
    opt.get_drop_fraction(np.int64(0), np.bool_(True))
    out['bad_anneal_error'] = None
  except ValueError as e:
    out['bad_anneal_error'] = str(e)
  return out


def gen_trajectory():
  """Per-step state of a full RigL run through the reference's minimize():
  Nesterov momentum inner optimizer, cosine drop fraction, zero noise."""
  traj = {}
  for tag, kw in [('rigl_mom', dict(inner='mom', acc_scale=0.0)),
                  ('rigl_mom_acc', dict(inner='mom', acc_scale=0.5)),
                  ('rigl_sgd', dict(inner='sgd'))]:
    opt, io, wv, mv, gs = _fc_problem(12, 20, 'rigl', 1, 17, 4, 0.4,
                                      anneal='cosine', lr=0.01, seed=3, **kw)
    _noise_inject(None)
    ws, ms, gss, moms, fracs = [wv.value.copy()], [mv.value.copy()], [0], [], []
    for _ in range(24):
      tf_shim.run_op(opt.minimize(None, gs))
      ws.append(wv.value.copy())
      ms.append(mv.value.copy())
      gss.append(int(gs.value))
      fracs.append(np.float32(opt.drop_fraction))
      if kw.get('inner') == 'mom':
        moms.append(io.get_slot(wv, 'momentum').value.copy())
    traj[tag + '__w'] = np.stack(ws)
    traj[tag + '__mask'] = np.stack(ms)
    traj[tag + '__gs'] = np.array(gss, np.int64)
    traj[tag + '__frac'] = np.array(fracs, np.float32)
    if moms:
      traj[tag + '__mom'] = np.stack(moms)
  return traj


def main():
  with open(os.path.join(HERE, 'mask_random.json'), 'w') as f:
    json.dump(gen_mask_random(), f, indent=1)
  with open(os.path.join(HERE, 'sparsities.json'), 'w') as f:
    json.dump(gen_sparsities(), f, indent=1)
  np.savez_compressed(os.path.join(HERE, 'update_cases.npz'),
                      **gen_update_cases())
  with open(os.path.join(HERE, 'schedule.json'), 'w') as f:
    json.dump(gen_schedule(), f, indent=1)
  np.savez_compressed(os.path.join(HERE, 'trajectory.npz'), **gen_trajectory())
  print('golden vectors written to', HERE)


if __name__ == '__main__':
  main()
