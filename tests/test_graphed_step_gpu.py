"""train.GraphedStep: ordinary iterations replayed from a captured HIP graph must leave the SAME bits as eager
execution -- weights, momentum slots, masks -- across a mask update that falls between replays."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _run(graphed, steps):
  from rigl_amd import sparse_optimizers as SO, sparse_utils, train, variables as V
  from rigl_amd.workloads import wide_resnet
  g = V.reset_default_graph(DEV)
  model = wide_resnet.WideResNet(g, depth=10, width=1)
  np.random.seed(0)
  sparse_utils.get_mask_init_fn(g.get_masks(), 'erdos_renyi_kernel', 0.8, {})()
  inner = train.MomentumOptimizer(0.05, 0.9, use_nesterov=True, graph=g)
  opt = SO.SparseRigLOptimizer(inner, 0, 1000, 5, drop_fraction=0.3, drop_fraction_anneal='cosine', noise_std=0.)
  gs = g.get_or_create_global_step()
  x, y = wide_resnet.synthetic_batch(32, DEV)
  loss_fn = lambda: model.loss(x, y)
  if graphed:
    st = train.GraphedStep(loss_fn, opt, gs, warmup=2)
    run = st
  else:
    st = None

    def run():
      loss = loss_fn()
      opt.minimize(loss, gs)
      return loss
  losses = []
  for _ in range(steps):
    losses.append(float(run().detach().float()))
  torch.cuda.synchronize()
  out = dict(W=g.W.cpu().numpy().copy(), A=inner._slot.cpu().numpy().copy(), B=g.BITS.cpu().numpy().copy(),
             gs=int(gs.value), losses=losses)
  if st is not None:
    out['replays'], out['eager'] = st.replays, st.eager_steps
  return out


def test_graph_replay_is_bit_identical_to_eager():
  steps = 19                      # calls 0, 6, 12, 18 are mask updates (period 5: an update does not advance the step)
  a = _run(False, steps)
  b = _run(True, steps)
  assert b['replays'] >= 8 and b['eager'] >= 4
  assert a['gs'] == b['gs'] == steps - 4
  np.testing.assert_array_equal(a['B'], b['B'])
  np.testing.assert_array_equal(a['W'].view(np.uint32), b['W'].view(np.uint32))
  np.testing.assert_array_equal(a['A'].view(np.uint32), b['A'].view(np.uint32))
  assert a['losses'] == b['losses']
