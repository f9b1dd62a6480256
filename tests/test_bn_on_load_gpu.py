"""relu(bn(x)) applied on the operand load of the conv that consumes it (round 6, VERDICT r5 item 2:
rigl_masked_conv2d_fwd_bnrelu, the BNL kernels of rowstream.hpp).  Same arithmetic at the same rounding points
(bf16(max(fma(x, scale, shift), 0)), then the conv): the output, the activated tensor it leaves for the backward and the
statistics parts must have the bits of the two separate calls (rigl_bn_fwd_stats + rigl_masked_conv2d_fwd_stats), and a
ResNet-50 step with the hand-over switched on the bits of one without.
Reference: batch_norm_relu between conv2 and conv3 of bottleneck_block_ (resnet_model.py:41-82, 456-470)."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

# n, h, cin, cout: the kernel variants <KC, TN> = <2,4> <4,4> <4,2> <8,2> <1,2> <2,2>, whole and ragged row counts
CASES = ((8, 28, 128, 512), (33, 14, 256, 1024), (5, 56, 256, 64), (8, 28, 512, 128), (3, 56, 64, 64), (7, 28, 128, 64),
         (128, 14, 256, 1024))


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
@pytest.mark.parametrize('n,h,ci,co', CASES)
def test_conv_with_the_apply_on_its_load_equals_the_two_calls(n, h, ci, co):
  from rigl_amd import ops
  ops.tune_set('rowstream', 2)
  ops.tune_set('bn_on_load', 1)
  try:
    d = ops.conv_desc(n, h, h, ci, co, 1, 1, 1, 0, 0, h, h)
    assert ops.conv_fwd_takes_bn_input(d), 'this shape was chosen to be taken'
    gen = torch.Generator(device=DEV).manual_seed(n * 1000 + ci)
    x = (torch.randn(n, h, h, ci, device=DEV, generator=gen) * 1.5 + 0.3).to(torch.bfloat16)
    w = (torch.randn(ci * co, device=DEV, generator=gen) * 0.05).to(torch.bfloat16)
    gamma = torch.rand(ci, device=DEV, generator=gen) + 0.5
    beta = torch.randn(ci, device=DEV, generator=gen) * 0.3
    rm0, rv0 = torch.zeros(ci, device=DEV), torch.ones(ci, device=DEV)
    rm1, rv1 = rm0.clone(), rv0.clone()
    a_ref, saved_ref = ops.bn_fwd(x, gamma, beta, rm0, rv0, 0.1, 1e-5, True, None)
    y_ref, p_ref = ops.conv_fwd(d, a_ref, w, stats=True)
    saved = ops.bn_statistics(x, gamma, beta, rm1, rv1, 0.1, 1e-5)
    a = torch.full_like(x, float('nan'))
    y, p = ops.conv_fwd_bnrelu(d, x, saved, w, a, stats=True)
    y_nostats = ops.conv_fwd_bnrelu(d, x, saved, w, torch.empty_like(x), stats=False)
    torch.cuda.synchronize()
    assert torch.equal(saved, saved_ref)
    assert torch.equal(a.view(torch.int16), a_ref.view(torch.int16)), 'activated tensor'
    assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16)), 'conv output'
    assert torch.equal(y_nostats.view(torch.int16), y_ref.view(torch.int16))
    assert torch.equal(p, p_ref), 'statistics parts'
    assert float(a_ref.float().abs().max()) > 0 and (a_ref == 0).any()      # the ReLU cut something and kept something
  finally:
    ops.tune_unset('rowstream')
    ops.tune_unset('bn_on_load')


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
def test_layers_that_do_not_take_it_say_so():
  from rigl_amd import ops
  ops.tune_set('bn_on_load', 1)
  try:
    for (n, h, ci, co, k) in ((128, 56, 64, 256, 1), (128, 14, 256, 256, 3), (128, 7, 512, 2048, 1)):
      d = ops.conv_desc(n, h, h, ci, co, k, k, 1, k // 2, k // 2, h, h)
      assert not ops.conv_fwd_takes_bn_input(d)
      x = torch.zeros(n, h, h, ci, device=DEV, dtype=torch.bfloat16)
      with pytest.raises(Exception, match="does not take the transform"):
        ops.conv_fwd_bnrelu(d, x, torch.zeros(4, ci, device=DEV), torch.zeros(k * k * ci * co, device=DEV, dtype=torch.bfloat16),
                            torch.empty_like(x))
    ops.tune_set('bn_on_load', 0)
    assert not ops.conv_fwd_takes_bn_input(ops.conv_desc(128, 14, 14, 256, 1024, 1, 1, 1, 0, 0, 14, 14))
  finally:
    ops.tune_unset('bn_on_load')


@pytest.mark.skipif(not torch.cuda.is_available(), reason='needs a GPU')
def test_resnet50_step_identical_with_the_apply_handed_to_conv3():
  from rigl_amd import ops, sparse_optimizers as SO, sparse_utils, train, variables as V
  from rigl_amd.workloads import nn as gnn, resnet50
  out, used = [], []
  for on in (True, False):
    g = V.reset_default_graph(DEV)
    model = resnet50.ResNet50(g, seed=0)
    np.random.seed(0)
    sparse_utils.get_mask_init_fn(g.get_masks(), 'erdos_renyi_kernel', 0.8, {})()
    for b in model.blocks:
      b.bn3.gamma.data.fill_(0.5)
    inner = train.MomentumOptimizer(0.05, 0.9, use_nesterov=True, graph=g)
    opt = SO.SparseRigLOptimizer(inner, 1, 25000, 100, drop_fraction=0.3, drop_fraction_anneal='cosine', noise_std=0.0)
    images, labels = resnet50.synthetic_batch(128, DEV, seed=5)    # (batch 128: the rows of the layers the body takes by default)
    calls = [0]
    fwd0 = ops.conv_fwd_bnrelu

    def counting(*a, **k):
      calls[0] += 1
      return fwd0(*a, **k)
    old = gnn._BN_ON_LOAD
    gnn._BN_ON_LOAD = on
    ops.tune_set('bn_on_load', 1)
    ops.conv_fwd_bnrelu = counting
    try:
      loss = model.loss(images, labels, label_smoothing=0.1)
      opt.compute_gradients(loss)
      torch.cuda.synchronize()
    finally:
      gnn._BN_ON_LOAD = old
      ops.conv_fwd_bnrelu = fwd0
      ops.tune_unset('bn_on_load')
    used.append(calls[0])
    stats = torch.cat([torch.cat([b.bn2.moving_mean, b.bn2.moving_variance]) for b in model.blocks])
    out.append((float(loss.detach()), g.G.detach().clone(), stats.clone()))
  assert used[0] == 10 and used[1] == 0, used        # conv3 of the 4 + 6 blocks of groups 2 and 3
  assert out[0][0] == out[1][0]
  assert torch.equal(out[0][1].view(torch.int32), out[1][1].view(torch.int32)), 'gradients differ'
  assert torch.equal(out[0][2], out[1][2]), 'moving statistics differ'
