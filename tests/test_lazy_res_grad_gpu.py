"""The shortcut gradient handed over unmasked (round 6): relu(bn3 + shortcut)'s backward gives the block's first conv the
output gradient itself + the ReLU bits its forward left, and the conv's dgrad epilogue masks on the fly
(rigl_masked_conv2d_bwd_masked, bwdslice.hpp) -- the masked copy is never written.  Same arithmetic at the same rounding
points: every gradient of a ResNet-50 step must have the bits of the eager hand-over (RIGL_LAZY_RES_GRAD=0).
Reference: the relu / add gradients of bottleneck_block_ through autodiff (resnet_model.py:497-501)."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
def test_resnet50_gradients_identical_with_and_without_the_masked_copy():
  from rigl_amd import ops, sparse_optimizers as SO, sparse_utils, train, variables as V
  from rigl_amd.workloads import nn as gnn, resnet50
  grads, used = [], []
  for lazy in (True, False):
    g = V.reset_default_graph(DEV)
    model = resnet50.ResNet50(g, seed=0)
    np.random.seed(0)
    sparse_utils.get_mask_init_fn(g.get_masks(), 'erdos_renyi_kernel', 0.8, {})()
    for b in model.blocks:                      # (bn3's gamma starts at zero: give the main branch a gradient to carry)
      b.bn3.gamma.data.fill_(0.5)
    inner = train.MomentumOptimizer(0.05, 0.9, use_nesterov=True, graph=g)
    opt = SO.SparseRigLOptimizer(inner, 1, 25000, 100, drop_fraction=0.3, drop_fraction_anneal='cosine', noise_std=0.0)
    images, labels = resnet50.synthetic_batch(32, DEV, seed=5)     # (batch 32: the group-2 / group-3 conv1 layers are on bwdslice)
    calls = [0]
    conv_bwd0 = ops.conv_bwd

    def counting(*a, **k):
      if k.get('addend_bits') is not None:
        calls[0] += 1
      return conv_bwd0(*a, **k)
    old = gnn._LAZY_RES_GRAD
    gnn._LAZY_RES_GRAD = lazy
    ops.conv_bwd = counting
    try:
      loss = model.loss(images, labels, label_smoothing=0.1)
      opt.compute_gradients(loss)
      torch.cuda.synchronize()
    finally:
      gnn._LAZY_RES_GRAD = old
      ops.conv_bwd = conv_bwd0
    assert not ops.LAZY_ADDEND_BITS, 'a lazy shortcut gradient was never taken by its conv'
    used.append(calls[0])
    grads.append((float(loss.detach()), g.G.detach().clone()))
  assert used[0] >= 6 and used[1] == 0, used        # groups 2 and 3: the non-first blocks' conv1 (3 + 5), if legal at this batch
  assert grads[0][0] == grads[1][0]
  assert torch.equal(grads[0][1].view(torch.int32), grads[1][1].view(torch.int32)), 'gradients differ with the masked copy skipped'
