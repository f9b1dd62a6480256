"""BASELINE.json's other configurations as parity-test cases: for each
workload one ordinary step (masked Nesterov update, K3) and one mask-update
step (K2) are compared BIT-EXACTLY with the oracle fed the device's own
gradients; sparsity bookkeeping is checked against the reference-derived
tables."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

from oracle import rigl_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _step_parity(g, loss_fn, opt, lr, mu, wd_by_kind, anneal, frac0, begin, end):
  from rigl_amd import variables as V
  inner = opt._optimizer
  gs = g.get_or_create_global_step()
  # ---- ordinary step -------------------------------------------------------
  gs.value = begin + 1
  opt._last_update_step = begin + 1            # just updated -> next call applies gradients
  gv = opt.compute_gradients(loss_fn())
  inner._ensure_slots()
  W0, G0, A0 = g.W.cpu().numpy(), g.G.cpu().numpy(), inner._slot.cpu().numpy()
  bits = g.BITS.cpu().numpy().view(np.uint32)
  opt.apply_gradients(gv, gs)
  assert gs.value == begin + 2
  for kind in (V.KIND_MASKED, V.KIND_DENSE, V.KIND_OTHER):
    b, e = g.seg[kind]
    if e <= b:
      continue
    if kind == V.KIND_MASKED:
      m = np.unpackbits(bits.view(np.uint8), bitorder='little')[b:e].astype(np.float32)
    else:
      m = np.ones(e - b, np.float32)
    gvar = O.masked_grad(G0[b:e], m, W0[b:e], wd_by_kind[kind])
    w_ref, a_ref = O.momentum_apply(W0[b:e], A0[b:e], gvar, lr, mu, nesterov=True)
    np.testing.assert_array_equal(g.W[b:e].cpu().numpy().view(np.uint32), w_ref.view(np.uint32))
    np.testing.assert_array_equal(inner._slot[b:e].cpu().numpy().view(np.uint32), a_ref.view(np.uint32))
  # ---- mask-update step ----------------------------------------------------
  opt._last_update_step = -10**6
  step = gs.value
  gv = opt.compute_gradients(loss_fn())
  layers = g.masked_layers()
  before = [(l.mask.numpy().copy(), l.weights.numpy().copy(), l.weights.grad.cpu().numpy().copy(),
             inner.get_slot(l.weights, 'momentum').cpu().numpy().copy()) for l in layers]
  ones0 = sum(int(b[0].sum()) for b in before)
  opt.apply_gradients(gv, gs)
  assert gs.value == step                       # F9: no increment on update iterations
  frac = O.get_drop_fraction(anneal, frac0, step, begin, end, True)
  assert np.float32(opt.drop_fraction) == np.float32(frac)
  for l, (m0, w0, g0, a0) in zip(layers, before):
    r = O.rigl_mask_update(m0, w0, g0, frac, momentum=a0)
    np.testing.assert_array_equal(l.mask.numpy(), r['mask'], err_msg=l.scope)
    np.testing.assert_array_equal(l.weights.numpy().view(np.uint32), r['weights'].view(np.uint32), err_msg=l.scope)
    np.testing.assert_array_equal(inner.get_slot(l.weights, 'momentum').cpu().numpy(), r['momentum'])
  assert sum(m.sum() for m in g.get_masks()) == ones0


def test_config1_mnist_mlp_90_uniform():
  """MNIST MLP 784-300-100-10, 90 % on layers 1-2, layer 3 dense (custom
  sparsity 0, mnist_train_eval.py:269-272), RigL dT=100 cosine, batch 100."""
  from rigl_amd import sparse_optimizers as SO, sparse_utils, train, variables as V
  from rigl_amd.workloads import mnist_mlp
  g = V.reset_default_graph(DEV)
  model = mnist_mlp.MnistMLP(g)
  np.random.seed(0)
  sparse_utils.get_mask_init_fn(g.get_masks(), 'random', 0.9, {'layer3': 0.0})()
  assert [m.sum() for m in g.get_masks()] == [23520, 3000, 1000]        # 27 520 non-zeros
  inner = train.MomentumOptimizer(0.2, 0.9, use_nesterov=True, graph=g)
  opt = SO.SparseRigLOptimizer(inner, 0, 50000, 100, drop_fraction=0.3, drop_fraction_anneal='cosine', noise_std=0.)
  x, y = mnist_mlp.synthetic_batch(100, DEV)
  from rigl_amd import variables as VV
  _step_parity(g, lambda: model.loss(x, y), opt, 0.2, 0.9,
               {VV.KIND_MASKED: 1e-4, VV.KIND_DENSE: 1e-4, VV.KIND_OTHER: 0.0}, 'cosine', 0.3, 0, 50000)


def test_config2_cifar_wrn22_erk80():
  """CIFAR WRN-22-1 ("ResNet-20"), ERK 0.8, dense stem, batch 128."""
  from rigl_amd import sparse_optimizers as SO, sparse_utils, train, variables as V
  from rigl_amd.workloads import wide_resnet, shapes as WS
  g = V.reset_default_graph(DEV)
  model = wide_resnet.WideResNet(g, depth=22, width=1)
  assert [m.name for m in g.get_masks()] == list(WS.wide_resnet_masks(22, 1).keys())
  np.random.seed(0)
  sparse_utils.get_mask_init_fn(g.get_masks(), 'erdos_renyi_kernel', 0.8, {})()
  assert sum(m.numel for m in g.get_masks()) == 270464
  assert sum(m.sum() for m in g.get_masks()) == 54109
  inner = train.MomentumOptimizer(0.1, 0.9, use_nesterov=True, graph=g)
  opt = SO.SparseRigLOptimizer(inner, 0, 75000, 100, drop_fraction=0.3, noise_std=0.)
  x, y = wide_resnet.synthetic_batch(128, DEV)
  _step_parity(g, lambda: model.loss(x, y), opt, 0.1, 0.9,
               {V.KIND_MASKED: 5e-4, V.KIND_DENSE: 5e-4, V.KIND_OTHER: 0.0}, 'constant', 0.3, 0, 75000)
  # and it trains
  opt._optimizer._lr = 0.02
  first = last = None
  for _ in range(30):
    loss = model.loss(x, y)
    opt.minimize(loss, g.get_or_create_global_step())
    v = float(loss.detach())
    first = v if first is None else first
    last = v
  assert np.isfinite(last) and last < first


def test_config4_resnet50_erk99_dense_stem():
  """ResNet-50 ERK 0.99 with a dense stem (README.md:17-20): the high-sparsity
  top-k stress -- the largest layer keeps 5 027 of 2 359 296 weights."""
  from rigl_amd import sparse_optimizers as SO, sparse_utils, train, variables as V
  from rigl_amd.workloads import resnet50
  g = V.reset_default_graph(DEV)
  model = resnet50.ResNet50(g, prune_first_layer=False, seed=0)
  np.random.seed(0)
  sparse_utils.get_mask_init_fn(g.get_masks(), 'erdos_renyi_kernel', 0.99, {})()
  assert len(g.get_masks()) == 53 and sum(m.numel for m in g.get_masks()) == 25493504
  assert sum(m.sum() for m in g.get_masks()) == 254978
  assert max(g.get_masks(), key=lambda m: m.numel).sum() == 5027
  inner = train.MomentumOptimizer(0.05, 0.9, use_nesterov=True, graph=g)
  opt = SO.SparseRigLOptimizer(inner, 0, 25000, 100, drop_fraction=0.3, drop_fraction_anneal='cosine', noise_std=0.)
  x, y = resnet50.synthetic_batch(8, DEV)
  _step_parity(g, lambda: model.loss(x, y, label_smoothing=0.1), opt, 0.05, 0.9,
               {V.KIND_MASKED: 1e-4, V.KIND_DENSE: 1e-4, V.KIND_OTHER: 0.0}, 'cosine', 0.3, 0, 25000)


def test_config5_mobilenet_v1_uniform90():
  """MobileNet-v1, --mask_init_method=random --end_sparsity=0.9 on the 13
  pointwise convs + final_dense; depthwise convs and the stem stay dense."""
  from rigl_amd import sparse_optimizers as SO, sparse_utils, train, variables as V
  from rigl_amd.workloads import mobilenet_v1, shapes as WS
  g = V.reset_default_graph(DEV)
  model = mobilenet_v1.MobileNetV1(g, seed=0)
  assert [m.name for m in g.get_masks()] == list(WS.mobilenet_v1_masks().keys())
  np.random.seed(0)
  sparse_utils.get_mask_init_fn(g.get_masks(), 'random', 0.9, {})()
  assert sum(m.numel for m in g.get_masks()) == 4163584
  assert sum(m.sum() for m in g.get_masks()) == 416365
  inner = train.MomentumOptimizer(0.05, 0.9, use_nesterov=True, graph=g)
  opt = SO.SparseRigLOptimizer(inner, 0, 25000, 100, drop_fraction=0.3, drop_fraction_anneal='cosine', noise_std=0.)
  x, y = mobilenet_v1.synthetic_batch(8, DEV)
  _step_parity(g, lambda: model.loss(x, y, label_smoothing=0.1), opt, 0.05, 0.9,
               {V.KIND_MASKED: 4e-5, V.KIND_DENSE: 4e-5, V.KIND_OTHER: 0.0}, 'cosine', 0.3, 0, 25000)
  # depthwise weights really train (dense gradient landed in the arena)
  dwv = [v for v in g.variables.values() if v.name.endswith('depthwise_weights:0')]
  assert len(dwv) == 13 and all(float(v.grad.abs().sum()) > 0 for v in dwv)


def test_config1_mnist_trajectory_every_update_bit_exact():
  """BASELINE config 1 over time: 330 iterations of the MNIST MLP (RigL dT = 100, cosine drop fraction), i.e. four mask
  updates at different drop fractions with training in between.  At EVERY update iteration the oracle is fed the
  device's own state (masks, weights, dense gradients, momentum) and must reproduce the new masks, the re-initialised
  weights and the momentum reset bit for bit; between updates the schedule's bookkeeping (no step increment on update
  iterations, last_update_step, drop fraction = fp32 cosine of the global step) is checked against the oracle."""
  from rigl_amd import sparse_optimizers as SO, sparse_utils, train, variables as V
  from rigl_amd.workloads import mnist_mlp
  g = V.reset_default_graph(DEV)
  model = mnist_mlp.MnistMLP(g)
  np.random.seed(3)
  sparse_utils.get_mask_init_fn(g.get_masks(), 'random', 0.9, {'layer3': 0.0})()
  inner = train.MomentumOptimizer(0.2, 0.9, use_nesterov=True, graph=g)
  begin, end, freq, f0 = 0, 300, 100, 0.3
  opt = SO.SparseRigLOptimizer(inner, begin, end, freq, drop_fraction=f0, drop_fraction_anneal='cosine', noise_std=0.)
  gs = g.get_or_create_global_step()
  layers = g.masked_layers()
  ones0 = [int(l.mask.numpy().sum()) for l in layers]
  updates = []
  for it in range(330):
    x, y = mnist_mlp.synthetic_batch(100, DEV, seed=1000 + it)
    step = int(gs.value)
    is_upd = (begin <= step <= end) and (opt._last_update_step + freq <= step)
    gv = opt.compute_gradients(model.loss(x, y))
    if is_upd:
      inner._ensure_slots()
      before = [(l.mask.numpy().copy(), l.weights.numpy().copy(), l.weights.grad.cpu().numpy().copy(),
                 inner.get_slot(l.weights, 'momentum').cpu().numpy().copy()) for l in layers]
    opt.apply_gradients(gv, gs)
    if not is_upd:
      assert int(gs.value) == step + 1
      continue
    assert int(gs.value) == step and opt._last_update_step == step          # F9: an update iteration takes no step
    frac = O.get_drop_fraction('cosine', f0, step, begin, end, True)
    assert np.float32(opt.drop_fraction) == np.float32(frac)
    for l, (m0, w0, g0, a0) in zip(layers, before):
      r = O.rigl_mask_update(m0, w0, g0, frac, momentum=a0)
      np.testing.assert_array_equal(l.mask.numpy(), r['mask'], err_msg='%s @ %d' % (l.scope, step))
      np.testing.assert_array_equal(l.weights.numpy().view(np.uint32), r['weights'].view(np.uint32))
      np.testing.assert_array_equal(inner.get_slot(l.weights, 'momentum').cpu().numpy().view(np.uint32),
                                    r['momentum'].view(np.uint32))
    updates.append((step, float(frac)))
  assert [u[0] for u in updates] == [0, 100, 200, 300]
  assert updates[0][1] > updates[1][1] > updates[2][1] > updates[3][1] >= 0.0     # cosine decay to zero at end_step
  assert [int(l.mask.numpy().sum()) for l in layers] == ones0                       # connections conserved throughout
