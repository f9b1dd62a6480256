"""The TF2-style front-end against golden vectors produced by EXECUTING the
reference's rigl/rigl_tf2/mask_updaters.py over the NumPy TF shim
(tests/golden/make_golden_tf2.py): the oracle and the host-side schedules on CPU,
the HIP-backed updaters on the GPU."""
import json
import os

import numpy as np
import pytest

from oracle import rigl_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, 'golden', 'tf2_updater.npz'))
S = json.load(open(os.path.join(HERE, 'golden', 'tf2_schedule.json')))
CASES = [(alg, f) for alg in ('rigl', 'rigl_inverted', 'set') for f in (0.3, 0.0, 1.0)]


def _scores(tag, alg, i):
  if alg == 'set':
    return G['%s__u%d' % (tag, i)]
  g = np.abs(G['%s__g%d' % (tag, i)])
  return g if alg == 'rigl' else -g


@pytest.mark.parametrize('alg,frac', CASES)
def test_oracle_reproduces_the_reference_tf2_update(alg, frac):
  tag = '%s_f%02d' % (alg, int(frac * 10))
  for i in range(3):
    w, m, a = G['%s__w%d' % (tag, i)], G['%s__m%d' % (tag, i)], G['%s__a%d' % (tag, i)]
    r = O.get_update(np.abs(m * w), _scores(tag, alg, i), m, w, np.float32(frac), momentum=a,
                     rigl_momentum_reset=False)
    np.testing.assert_array_equal(r['mask'], G['%s__m%d_new' % (tag, i)])
    np.testing.assert_array_equal(r['weights'], G['%s__w%d_new' % (tag, i)])
    np.testing.assert_array_equal(r['momentum'], G['%s__a%d_new' % (tag, i)])


def test_oracle_reproduces_prune_and_reinit():
  for i in range(3):
    w, m = G['prune__w%d' % i], G['prune__m%d' % i]
    sd = np.abs(m * w).reshape(-1)
    n_ones = np.int32(m.sum())
    n_keep = n_ones - np.int32(np.float32(n_ones) * np.float32(0.4))
    want = np.zeros(sd.size, np.float32)
    want[O.topk_order(sd)[:n_keep]] = 1
    np.testing.assert_array_equal(want.reshape(m.shape), G['prune__m%d_new' % i])
    np.testing.assert_array_equal(w, G['prune__w%d_new' % i])
  r = O.get_update(G['reinit__sd'], G['reinit__sg'], G['reinit__m'], G['reinit__w'], np.float32(0.5),
                   momentum=G['reinit__a'], reinit_when_same=True, rigl_momentum_reset=False)
  np.testing.assert_array_equal(r['mask'], G['reinit__m_new'])
  np.testing.assert_array_equal(r['weights'], G['reinit__w_new'])
  np.testing.assert_array_equal(r['momentum'], G['reinit__a_new'])


def test_schedules_match_the_reference_bit_for_bit():
  from rigl.rigl_tf2 import mask_updaters as MU

  class U:
    def update_masks(self, f):
      pass

    def prune_masks(self, f):
      pass
  for row in S['gate']:
    s = MU.ConstantUpdateSchedule(U(), 0.3, row['update_freq'], row['last_update_step'])
    assert [k for k in range(0, 1201, 50) if s.is_update_iter(k)] == row['update_steps']
  c = MU.CosineUpdateSchedule(U(), 0.3, 100, 1000)
  for step, hx in S['cosine']:
    assert float(np.float32(c.get_drop_fraction(step))).hex() == hx, step

  class Opt:
    def __init__(self, lr):
      self.lr = lr
  sched = MU.ScaledLRUpdateSchedule(U(), 0.3, 10, -1, Opt(lambda step: 0.1 * (0.5 ** (step // 100))))
  for step, hx in S['lr']:
    assert float(np.float32(sched.get_drop_fraction(step))).hex() == hx, step
  k = MU.ConstantUpdateSchedule(U(), 0.2, 10, -1)
  assert [float(np.float32(k.get_drop_fraction(s_))).hex() for s_ in (0, 10, 12345)] == S['constant']


def _graph(dev):
  from rigl_amd import pruning_layers as PL, train, variables as V
  g = V.reset_default_graph(dev)
  PL.MaskedDense(g, 'layer0', 24, 16, use_bias=False, sparsity_technique='threshold')
  PL.MaskedConv2d(g, 'layer1', 8, 16, (3, 3), sparsity_technique='threshold')
  PL.MaskedDense(g, 'layer2', 8, 8, use_bias=False, sparsity_technique='threshold')     # the reference's (64,) tensor
  g.finalize()
  opt = train.MomentumOptimizer(0.1, 0.9, graph=g)
  opt._ensure_slots()
  return g, opt


def _load(g, opt, tag, with_slots=True):
  import torch
  for i, l in enumerate(g.masked_layers()):
    w = G['%s__w%d' % (tag, i)].reshape(l.weights.shape)
    with torch.no_grad():
      l.weights.data.copy_(torch.from_numpy(w))
      if with_slots:
        opt.get_slot(l.weights, 'momentum').copy_(torch.from_numpy(G['%s__a%d' % (tag, i)].reshape(l.weights.shape)))
    l.mask.assign(G['%s__m%d' % (tag, i)].reshape(l.weights.shape))


@pytest.mark.gpu
@pytest.mark.parametrize('alg,frac', CASES)
def test_hip_updaters_reproduce_the_reference_tf2_update(alg, frac):
  import torch
  from rigl.rigl_tf2 import mask_updaters as MU
  g, opt = _graph('cuda:0')
  tag = '%s_f%02d' % (alg, int(frac * 10))
  _load(g, opt, tag)
  layers = g.masked_layers()
  cls = {'rigl': MU.RigL, 'rigl_inverted': MU.RigLInverted, 'set': MU.SET}[alg]
  up = cls(g, opt)
  if alg == 'set':
    uni = [torch.from_numpy(G['%s__u%d' % (tag, i)]).to('cuda:0') for i in range(3)]
    up.get_grow_scores = lambda all_vars, all_masks: uni
  else:
    for i, l in enumerate(layers):
      l.weights.grad.copy_(torch.from_numpy(G['%s__g%d' % (tag, i)].reshape(l.weights.shape)))
    up._get_gradients = lambda all_vars: [v.grad for v in all_vars]
  up.update_masks(np.float32(frac))               # called directly, like the generator (f = 0 still re-ranks an active exact zero)
  for i, l in enumerate(layers):
    np.testing.assert_array_equal(l.mask.numpy().reshape(-1), G['%s__m%d_new' % (tag, i)].reshape(-1), err_msg='mask %d' % i)
    np.testing.assert_array_equal(l.weights.data.cpu().numpy().reshape(-1), G['%s__w%d_new' % (tag, i)].reshape(-1))
    np.testing.assert_array_equal(opt.get_slot(l.weights, 'momentum').cpu().numpy().reshape(-1),
                                  G['%s__a%d_new' % (tag, i)].reshape(-1))


@pytest.mark.gpu
def test_hip_prune_and_reinit_reproduce_the_reference():
  import torch
  from rigl.rigl_tf2 import mask_updaters as MU
  g, opt = _graph('cuda:0')
  _load(g, opt, 'prune', with_slots=False)
  MU.RigL(g, opt).prune_masks(np.float32(0.4))
  for i, l in enumerate(g.masked_layers()):
    np.testing.assert_array_equal(l.mask.numpy().reshape(-1), G['prune__m%d_new' % i].reshape(-1))
    np.testing.assert_array_equal(l.weights.data.cpu().numpy().reshape(-1), G['prune__w%d_new' % i].reshape(-1))
  l = g.masked_layers()[0]
  with torch.no_grad():
    l.weights.data.copy_(torch.from_numpy(G['reinit__w']))
    opt.get_slot(l.weights, 'momentum').copy_(torch.from_numpy(G['reinit__a']))
  l.mask.assign(G['reinit__m'])
  MU.MaskUpdater(g, opt).generic_mask_update(l.mask, l.weights, torch.from_numpy(G['reinit__sd']).to('cuda:0'),
                                             torch.from_numpy(G['reinit__sg']).to('cuda:0'), np.float32(0.5),
                                             reinit_when_same=True)
  np.testing.assert_array_equal(l.mask.numpy(), G['reinit__m_new'])
  np.testing.assert_array_equal(l.weights.data.cpu().numpy(), G['reinit__w_new'])
  np.testing.assert_array_equal(opt.get_slot(l.weights, 'momentum').cpu().numpy(), G['reinit__a_new'])
