"""TF2-style front-end (rigl_tf2/mask_updaters.py): schedules on the host, the
updaters through the fused K2 kernel, compared bit-exactly with the oracle's
_get_update_op restatement (which is pinned to the executed reference; the TF2
generic_mask_update is the same algorithm with zero grow values and zeroed
slots -- Appendix B of SURVEY.md)."""
import math

import numpy as np
import pytest

torch = pytest.importorskip('torch')

from oracle import rigl_oracle as O  # noqa: E402
from oracle import tf_random as T  # noqa: E402


class _FakeUpdater:
  def __init__(self):
    self.calls = []

  def update_masks(self, f):
    self.calls.append(('update', float(f)))

  def prune_masks(self, f):
    self.calls.append(('prune', float(f)))

  def set_validation_data(self, x, y):
    self.calls.append(('val', x, y))


def test_update_gate_matches_the_reference_rules():
  from rigl.rigl_tf2 import mask_updaters as MU
  s = MU.ConstantUpdateSchedule(_FakeUpdater(), 0.3, 100, -1)
  assert [k for k in range(0, 501, 50) if s.is_update_iter(k)] == [0, 100, 200, 300, 400, 500]   # < 0: forever
  s = MU.ConstantUpdateSchedule(_FakeUpdater(), 0.3, 100, 0)
  assert not any(s.is_update_iter(k) for k in range(0, 500, 100))                               # 0: never
  s = MU.ConstantUpdateSchedule(_FakeUpdater(), 0.3, 100, 250)
  assert [k for k in range(0, 501, 100) if s.is_update_iter(k)] == [0, 100, 200]
  with pytest.raises(AssertionError):
    s.is_update_iter(-1)
  with pytest.raises(AssertionError):
    s.update(150)
  s.update(150, check_update_iter=False)
  assert s._mask_updater.calls == [('update', pytest.approx(0.3))]


def test_schedules():
  from rigl.rigl_tf2 import mask_updaters as MU
  up = _FakeUpdater()
  c = MU.CosineUpdateSchedule(up, 0.3, 100, 1000)
  for step in (0, 100, 500, 1000, 1500):
    want = np.float32(0.3) * np.float32(0.5 * (1 + math.cos(math.pi * min(step, 1000) / 1000)))
    assert abs(float(c.get_drop_fraction(step)) - float(want)) < 1e-7
  assert float(c.get_drop_fraction(1000)) < 1e-7
  c.update(1000)                       # fraction 0 -> no update (tf.cond(last_drop_fraction > 0))
  assert up.calls == []
  c.update(100)
  assert up.calls[-1][0] == 'update' and abs(up.calls[-1][1] - float(c.get_drop_fraction(100))) < 1e-9
  c.prune(0.5)
  assert up.calls[-1] == ('prune', 0.5) and c.last_drop_fraction == 0.5

  class Opt:
    def __init__(self, lr):
      self.lr = lr
  sched = MU.ScaledLRUpdateSchedule(up, 0.3, 10, -1, Opt(lambda step: 0.1 * (0.5 ** (step // 100))))
  assert abs(float(sched.get_drop_fraction(0)) - 0.3) < 1e-7
  assert abs(float(sched.get_drop_fraction(250)) - 0.075) < 1e-7
  assert abs(float(MU.ScaledLRUpdateSchedule(up, 0.2, 10, -1, Opt(0.05)).get_drop_fraction(77)) - 0.2) < 1e-7
  assert MU.get_mask_updater(None, None, None, update_alg='') is None
  with pytest.raises(ValueError):
    MU.get_mask_updater(None, Opt(0.1), None, update_alg='nope')


def _mlp(dev, momentum=True):
  from rigl_amd import pruning_layers as PL, sparse_utils, train, variables as V
  g = V.reset_default_graph(dev)
  l1 = PL.MaskedDense(g, 'fc1', 48, 64, use_bias=False, sparsity_technique='threshold')
  l2 = PL.MaskedDense(g, 'fc2', 64, 16, use_bias=False, sparsity_technique='threshold')
  g.finalize()
  np.random.seed(3)
  sparse_utils.get_mask_init_fn(g.get_masks(), 'random', 0.7, {})()
  opt = train.MomentumOptimizer(0.1, 0.9, graph=g) if momentum else train.GradientDescentOptimizer(0.1, graph=g)
  x = torch.randn(32, 48, device=dev).to(torch.bfloat16)
  y = torch.randint(0, 16, (32,), device=dev)

  def loss_fn(vx, vy):
    return torch.nn.functional.cross_entropy(l2(torch.relu(l1(vx))).float(), vy)
  return g, opt, loss_fn, x, y


def _snapshot(g, opt):
  out = []
  for l in g.masked_layers():
    slot = opt.get_slot(l.weights, 'momentum').cpu().numpy().copy() if opt.get_slot_names() else None
    out.append((l.mask.numpy().copy(), l.weights.data.cpu().numpy().copy(), slot))
  return out


@pytest.mark.gpu
@pytest.mark.parametrize('alg', ['rigl', 'rigl_inverted', 'set'])
def test_updaters_match_the_oracle(alg):
  from rigl.rigl_tf2 import mask_updaters as MU
  from rigl_amd import pyhash
  g, opt, loss_fn, x, y = _mlp('cuda:0')
  opt._ensure_slots()
  opt._slot.normal_()                                     # non-trivial momentum to see the reset
  sched = MU.get_mask_updater(g, opt, loss_fn, update_alg=alg, schedule_alg='constant', update_freq=10,
                              init_drop_fraction=0.25, last_update_step=-1)
  sched.set_validation_data(x, y)
  g.get_or_create_global_step().value = 30
  before = _snapshot(g, opt)
  ones = [int(b[0].sum()) for b in before]
  sched.update(30)
  for l, (m0, w0, a0) in zip(g.masked_layers(), before):
    grad = l.weights.grad.cpu().numpy()
    sd = np.abs(m0 * w0)
    if alg == 'set':
      s0, s1 = T.tf_seed_pair(0, pyhash.name_hash(l.weights.name + 'grow'), 30)
      sg = T.stateless_random_uniform(w0.size, s0, s1).reshape(w0.shape)
    else:
      sg = np.abs(grad) if alg == 'rigl' else -np.abs(grad)
    r = O.get_update(sd, sg, m0, w0, 0.25, momentum=a0, rigl_momentum_reset=False)
    np.testing.assert_array_equal(l.mask.numpy(), r['mask'], err_msg=l.scope)
    np.testing.assert_array_equal(l.weights.data.cpu().numpy().view(np.uint32), r['weights'].view(np.uint32))
    np.testing.assert_array_equal(opt.get_slot(l.weights, 'momentum').cpu().numpy(), r['momentum'])
    assert r['n_prune'] > 0
  assert [m.sum() for m in g.get_masks()] == ones          # connections conserved


@pytest.mark.gpu
def test_prune_masks_keeps_the_largest_magnitudes():
  from rigl.rigl_tf2 import mask_updaters as MU
  g, opt, loss_fn, x, y = _mlp('cuda:0', momentum=False)
  sched = MU.get_mask_updater(g, opt, loss_fn, update_alg='set', schedule_alg='constant')
  before = _snapshot(g, opt)
  sched.prune(0.4)
  for l, (m0, w0, _) in zip(g.masked_layers(), before):
    n_ones = int(m0.sum())
    n_keep = n_ones - int(np.float32(n_ones) * np.float32(0.4))
    sd = np.abs(m0 * w0).reshape(-1)
    want = np.zeros(sd.size, np.float32)
    want[O.topk_order(sd)[:n_keep]] = 1
    np.testing.assert_array_equal(l.mask.numpy().reshape(-1), want)
    np.testing.assert_array_equal(l.weights.data.cpu().numpy(), w0)     # weights untouched


@pytest.mark.gpu
def test_generic_mask_update_with_explicit_scores_and_reinit():
  from rigl.rigl_tf2 import mask_updaters as MU
  g, opt, loss_fn, x, y = _mlp('cuda:0')
  opt._ensure_slots()
  opt._slot.fill_(2.0)
  up = MU.MaskUpdater(g, opt)
  l = g.masked_layers()[0]
  m0, w0 = l.mask.numpy().copy(), l.weights.data.cpu().numpy().copy()
  gen = torch.Generator(device='cuda:0').manual_seed(5)
  sd = torch.rand(w0.shape, generator=gen, device='cuda:0')
  sg = torch.rand(w0.shape, generator=gen, device='cuda:0')
  up.generic_mask_update(l.mask, l.weights, sd, sg, 0.5, reinit_when_same=True)
  r = O.get_update(sd.cpu().numpy(), sg.cpu().numpy(), m0, w0, 0.5, momentum=np.full(w0.shape, 2.0, np.float32),
                   reinit_when_same=True, rigl_momentum_reset=False)
  np.testing.assert_array_equal(l.mask.numpy(), r['mask'])
  np.testing.assert_array_equal(l.weights.data.cpu().numpy(), r['weights'])
  np.testing.assert_array_equal(opt.get_slot(l.weights, 'momentum').cpu().numpy(), r['momentum'])
