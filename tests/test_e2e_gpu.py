"""End-to-end parity on the GPU through the reference-shaped API.

Part 1 re-states rigl/sparse_optimizers_test.py (same toy graphs, same
assertions) on the HIP-backed optimizers.  Part 2 checks one real ResNet-50
RigL step: forward/backward against the fp32 CPU restatement of the reference
model (oracle/resnet_cpu.py), then the optimizer update and the mask update
bit-exactly against the oracle fed with the SAME gradients."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

from oracle import rigl_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _setup(kind, n_inp, n_out, drop_frac, start_iter=1, end_iter=4, freq_iter=2, lr=0.1, x_ones=False,
           scale_by_step=False, **kw):
  """The `_setup_graph` of the reference tests (sparse_optimizers_test.py:40-69,
  :299-328): one masked FC layer, loss = mean(y) | sum(y * arange * step)."""
  from rigl_amd import pruning, pruning_layers as PL, sparse_optimizers as SO, train, variables as V
  g = V.reset_default_graph(DEV)
  inner = train.GradientDescentOptimizer(lr)
  cls = {'set': SO.SparseSETOptimizer, 'static': SO.SparseStaticOptimizer,
         'momentum': SO.SparseMomentumOptimizer, 'rigl': SO.SparseRigLOptimizer}[kind]
  opt = cls(inner, start_iter, end_iter, freq_iter, drop_fraction=drop_frac, **kw)
  gs = train.get_or_create_global_step()
  rs = np.random.RandomState(0)

  def loss_fn():
    x = torch.ones(1, n_inp) if x_ones else torch.from_numpy(rs.rand(1, n_inp).astype(np.float32))
    y = PL.sparse_fully_connected(x.to(DEV, torch.bfloat16), n_out, sparsity_technique='threshold',
                                  use_bias=True, name='fully_connected').float()
    if scale_by_step:
      scale = torch.arange(n_out, device=DEV, dtype=torch.float32) * float(gs.value)
      return (y * scale).sum(), scale
    if x_ones:
      return (y * torch.arange(n_out, device=DEV, dtype=torch.float32)).sum(), None
    return y.mean(), None

  loss_fn()                                    # creates the variables
  weight, mask = pruning.get_weights()[0], pruning.get_masks()[0]
  mask.assign(np.random.RandomState(1).choice([0, 1], size=(n_inp, n_out), p=[0.5, 0.5]).astype(np.float32))
  return opt, loss_fn, mask, weight, gs


@pytest.mark.parametrize('n_inp,n_out,drop_frac', [(15, 25, 0.5), (15, 25, 0.2), (3, 5, 0.2)])
def test_set_mask_non_update_iterations(n_inp, n_out, drop_frac):
  # sparse_optimizers_test.py:71-92
  opt, loss_fn, mask, _, gs = _setup('set', n_inp, n_out, drop_frac)
  for i in range(1, 6):
    m0 = mask.numpy()
    opt.minimize(loss_fn()[0], gs)
    if i not in (1, 3):
      np.testing.assert_array_equal(m0, mask.numpy())


@pytest.mark.parametrize('n_inp,n_out,drop_frac', [(15, 25, 0.5), (15, 25, 0.7), (30, 10, 0.9)])
def test_set_update_iterations(n_inp, n_out, drop_frac):
  # :94-118
  opt, loss_fn, mask, _, gs = _setup('set', n_inp, n_out, drop_frac)
  for i in range(1, 5):
    m0 = mask.numpy()
    opt.minimize(loss_fn()[0], gs)
    m1 = mask.numpy()
    if i in (1, 3):
      assert m0.sum() == m1.sum()
      assert not np.array_equal(m0, m1)


@pytest.mark.parametrize('start_iter,end_iter,freq_iter', [(3, 7, 2), (1, 5, 3), (0, 4, 1)])
def test_set_no_drop(start_iter, end_iter, freq_iter):
  # :120-139
  opt, loss_fn, mask, _, gs = _setup('set', 3, 5, 0, start_iter, end_iter, freq_iter)
  for _ in range(end_iter + 2):
    m0 = mask.numpy()
    opt.minimize(loss_fn()[0], gs)
    np.testing.assert_array_equal(m0, mask.numpy())


def test_set_new_connection_zero_init():
  # :141-156
  opt, loss_fn, mask, weight, gs = _setup('set', 3, 5, 0.5, 0, 4, 1)
  for _ in range(5):
    m0 = mask.numpy()
    opt.minimize(loss_fn()[0], gs)
    m1, w1 = mask.numpy(), weight.numpy()
    assert np.all(w1[np.logical_and(m0 == 0, m1 == 1)] == 0)


def test_grow_init_initial_dist_is_a_stateless_permutation():
  """grow_init='initial_dist_<d>' (sparse_optimizers_base.py:372-380): new connections start from a random
  permutation of the variable's initial values / d.  The reference's tf.random_shuffle is a stateful op, so the
  pinned properties are: a permutation (same multiset), reproducible from (seed offset, global step) alone --
  replicas agree with no traffic --, different at another step, and the grown weights of an update are drawn from it."""
  opt, loss_fn, mask, weight, gs = _setup('rigl', 15, 25, 0.5, 0, 4, 1)
  opt._grow_init = 'initial_dist_2'
  init = torch.randn(15, 25, device=DEV)
  weight.initial_value = init
  gs.value = 3
  opt._global_step = gs
  t1 = opt.get_grow_tensor(weight, 'initial_dist_2')
  t1b = opt.get_grow_tensor(weight, 'initial_dist_2')
  assert torch.equal(t1, t1b)
  assert torch.equal(torch.sort(t1.reshape(-1)).values, torch.sort((init / 2).reshape(-1)).values)
  assert not torch.equal(t1, init / 2)
  gs.value = 4
  assert not torch.equal(opt.get_grow_tensor(weight, 'initial_dist_2'), t1)
  gs.value = 0
  m0 = mask.numpy()
  opt.minimize(loss_fn()[0], gs)                 # step 0 is a mask update (begin_step = 0)
  grown = np.logical_and(m0 == 0, mask.numpy() == 1)
  assert grown.any()
  expect = opt.get_grow_tensor(weight, 'initial_dist_2').cpu().numpy()   # same (seed, step) -> same tensor
  np.testing.assert_array_equal(weight.numpy()[grown], expect[grown])


@pytest.mark.parametrize('n_inp,n_out,drop_frac', [(15, 25, 0.5), (15, 25, 0.2), (3, 5, 0.2)])
def test_static_mask_never_changes(n_inp, n_out, drop_frac):
  # :225-244
  opt, loss_fn, mask, _, gs = _setup('static', n_inp, n_out, drop_frac)
  for _ in range(5):
    m0 = mask.numpy()
    opt.minimize(loss_fn()[0], gs)
    np.testing.assert_array_equal(m0, mask.numpy())


@pytest.mark.parametrize('n_inp,n_out,momentum', [(3, 4, 0.5), (5, 2, 0.), (2, 5, 1.)])
def test_momentum_ema_update(n_inp, n_out, momentum):
  # :275-294
  opt, loss_fn, _, weight, gs = _setup('momentum', n_inp, n_out, 0.5, x_ones=True, momentum=momentum)
  cur = np.zeros((n_inp, n_out), np.float32)
  for _ in range(6):
    opt.minimize(loss_fn()[0], gs)
    cur = (cur * np.float32(momentum) + np.float32(1 - momentum) * np.arange(n_out, dtype=np.float32)).astype(np.float32)
    np.testing.assert_allclose(opt.ema_average(weight).cpu().numpy(), cur, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('n_inp,n_out', [(3, 4), (5, 2), (2, 5)])
def test_rigl_masked_gradient_is_dense(n_inp, n_out):
  # :330-347 -- _weight2masked_grads equals the dense analytic gradient
  opt, loss_fn, _, weight, gs = _setup('rigl', n_inp, n_out, 0., 0, 3, 1, lr=1e-3, x_ones=True, scale_by_step=True)
  for i in range(6):
    loss, scale = loss_fn()
    opt.minimize(loss, gs)
    if i % 2 == 0:
      expected = scale.unsqueeze(0).expand(n_inp, n_out).cpu().numpy()
      np.testing.assert_array_equal(opt._weight2masked_grads[weight.name].cpu().numpy(), expected)
      from rigl_amd.sparse_optimizers import get_grow_grads
      np.testing.assert_array_equal(get_grow_grads(opt)[0].cpu().numpy(), expected)


@pytest.mark.parametrize('start_iter,end_iter,freq_iter,is_incremented', [
    (3, 7, 2, [1, 1, 1, 0, 1, 1, 0, 1, 1, 0, 1]),
    (1, 5, 3, [1, 0, 1, 1, 1, 0, 1, 1, 1, 1, 1]),
    (0, 4, 1, [0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 1])])
def test_rigl_apply_gradients_golden(start_iter, end_iter, freq_iter, is_incremented):
  # :349-367
  opt, loss_fn, _, _, gs = _setup('rigl', 3, 5, .5, start_iter, end_iter, freq_iter, lr=1e-3, x_ones=True,
                                  scale_by_step=True)
  for one in is_incremented:
    before = gs.value
    opt.minimize(loss_fn()[0], gs)
    assert gs.value == before + one


# ---------------------------------------------------------------------------- SNIP / DNW
def _setup_oneshot(kind, default_sparsity, n_inp, n_out, l2=0.0):
  """sparse_optimizers_test.py:372-402 / :472-507: x = 1..n_inp, loss = sum(y * scale)."""
  from rigl_amd import pruning, pruning_layers as PL, sparse_optimizers as SO, train, variables as V
  g = V.reset_default_graph(DEV)
  inner = train.GradientDescentOptimizer(1e-3)
  cls = SO.SparseSnipOptimizer if kind == 'snip' else SO.SparseDNWOptimizer
  opt = cls(inner, default_sparsity, 'random', custom_sparsity_map={})
  inp = np.arange(1, n_inp + 1).astype(np.float32)
  scale = (np.random.RandomState(n_inp * 7 + n_out).uniform(size=(n_out,)) - 0.5).astype(np.float32)
  scale = torch.from_numpy(scale).to(torch.bfloat16).float().numpy()       # exactly representable downstream
  expected_grads = np.outer(inp, scale)
  gs = train.get_or_create_global_step()

  def loss_fn():
    x = torch.from_numpy(inp).reshape(1, n_inp).to(DEV, torch.bfloat16)
    y = PL.sparse_fully_connected(x, n_out, sparsity_technique='threshold', name='fully_connected',
                                  kernel_regularizer=PL.l2_regularizer(l2) if l2 else None).float()
    return (y * torch.from_numpy(scale).to(DEV)).sum()

  loss_fn()
  return opt, loss_fn, expected_grads, pruning.get_masks()[0], pruning.get_weights()[0], gs


def test_snip_score_includes_the_l2_gradient():
  """The reference scores |g * v| with g = d(loss incl. the regulariser)/d(variable) (sparse_optimizers.py:299-301); here
  the l2 gradient normally lives inside the update kernel, so SNIP has to add it back."""
  from rigl_amd import sparse_utils
  l2 = 50.0                                                        # large enough to reorder the scores
  opt, loss_fn, expected_grads, mask, weights, gs = _setup_oneshot('snip', 0.5, 6, 5, l2=l2)
  w0 = weights.numpy().copy()
  assert opt.minimize(loss_fn(), gs)
  m = mask.numpy()
  scores = np.abs((expected_grads + np.float32(l2) * w0) * w0)
  assert scores[m == 0].max() <= scores[m == 1].min()
  plain = np.abs(expected_grads * w0)
  keep = m.size - sparse_utils.get_n_zeros(m.size, 0.5)
  assert not np.array_equal(np.sort(np.argsort(-plain.reshape(-1), kind='stable')[:keep]), np.flatnonzero(m.reshape(-1)))


def test_snip_masks_do_not_depend_on_the_number_of_replicas():
  """Under data parallelism the arena holds the SUM of the replicas' data gradients and 1 / world rides in grad_scale: the
  SNIP score must scale the sum back before adding wd * W.  Two identical replicas are emulated by doubling the
  loss (= the summed gradient) and a GradSync stand-in with grad_scale 1/2; the masks must be those of the one-GPU run."""
  l2 = 50.0
  opt, loss_fn, _, mask, weights, gs = _setup_oneshot('snip', 0.5, 6, 5, l2=l2)
  w0 = weights.data.clone()
  assert opt.minimize(loss_fn(), gs)
  m1 = mask.numpy().copy()

  class _Sync:                       # what SparseSnipOptimizer and train.Optimizer read of a rigl_amd.dist.GradSync
    enabled, world, grad_scale = True, 2, 0.5

    def all_reduce(self, graph):     # the "sum over replicas" already happened (doubled loss)
      del graph

  opt2, loss_fn2, _, mask2, weights2, gs2 = _setup_oneshot('snip', 0.5, 6, 5, l2=l2)
  weights2.data.copy_(w0)                                                   # the same initial weights as the one-GPU run
  weights2.graph.shadows_dirty = True
  opt2._optimizer._grad_sync = _Sync()                                      # pylint: disable=protected-access
  assert opt2.minimize(2.0 * loss_fn2(), gs2)
  np.testing.assert_array_equal(mask2.numpy(), m1)


def test_dnw_masked_kernels_get_no_l2_gradient():
  """DNW differentiates w.r.t. the masked_weights tensors (sparse_optimizers.py:375-386): the regulariser on the raw
  variables does not reach them, so the update is w -= lr * dense_grad exactly, whatever the l2 scale."""
  opt, loss_fn, expected_grads, mask, weights, gs = _setup_oneshot('dnw', 0.5, 6, 5, l2=0.3)
  w0 = weights.numpy().copy()
  gv = opt.compute_gradients(loss_fn())
  opt.apply_gradients(gv, gs)
  np.testing.assert_array_equal(weights.numpy(), (w0 - np.float32(1e-3) * expected_grads.astype(np.float32)).astype(np.float32))


@pytest.mark.parametrize('n_inp,n_out,s', [(3, 4, 0.5), (5, 3, 0.8), (8, 5, 0.8)])
def test_snip(n_inp, n_out, s):
  # sparse_optimizers_test.py:404-468
  from rigl_amd import sparse_utils
  opt, loss_fn, expected_grads, mask, weights, gs = _setup_oneshot('snip', s, n_inp, n_out)
  assert mask.numpy().sum() == mask.numel                       # testInitialMaskIsDense
  w0 = weights.numpy().copy()
  is_snip = opt.minimize(loss_fn(), gs)
  assert is_snip and gs.value == 0
  m = mask.numpy()
  assert m.size - m.sum() == sparse_utils.get_n_zeros(m.size, s)  # testSnipSparsity
  scores = np.abs(expected_grads * w0)                          # testGradientUsed
  assert scores[m == 0].max() <= scores[m == 1].min()
  np.testing.assert_array_equal(weights.numpy(), w0)            # no gradient applied on the snip step
  for i in range(3):                                            # testAfterSnipTraining
    assert gs.value == i
    assert opt.minimize(loss_fn(), gs) is False and opt.is_snipped
    np.testing.assert_array_equal(mask.numpy(), m)


@pytest.mark.parametrize('n_inp,n_out,s', [(3, 4, 0.5), (5, 3, 0.8), (8, 5, 0.8)])
def test_dnw(n_inp, n_out, s):
  # sparse_optimizers_test.py:509-586
  from rigl_amd import sparse_utils
  opt, loss_fn, expected_grads, mask, weights, gs = _setup_oneshot('dnw', s, n_inp, n_out)
  gv = opt.compute_gradients(loss_fn())
  grad = [g for g, v in gv if v is weights][0]
  np.testing.assert_allclose(grad.cpu().numpy(), expected_grads, rtol=1e-6)   # testGradientIsDense
  opt.apply_gradients(gv, gs)
  for _ in range(5):                                            # testDNWUpdates / testSparsityAfterDNWUpdates
    m, w = mask.numpy(), weights.numpy()
    assert m.size - m.sum() == sparse_utils.get_n_zeros(m.size, s)
    assert np.abs(w[m == 0]).max() <= np.abs(w[m == 1]).min()
    opt.minimize(loss_fn(), gs)
  # the dense gradient moved masked-out weights too (DNW's point)
  assert np.all(weights.numpy() != 0)


# ---------------------------------------------------------------------------- ResNet-50
@pytest.fixture(scope='module')
def rn50():
  from rigl_amd import sparse_optimizers as SO, sparse_utils, train, variables as V
  from rigl_amd.workloads import resnet50
  g = V.reset_default_graph(DEV)
  model = resnet50.ResNet50(g, seed=0)
  np.random.seed(0)
  sparse_utils.get_mask_init_fn(g.get_masks(), 'erdos_renyi_kernel', 0.8, {})()
  inner = train.MomentumOptimizer(0.05, 0.9, use_nesterov=True, graph=g)
  opt = SO.SparseRigLOptimizer(inner, 1, 25000, 100, drop_fraction=0.3, drop_fraction_anneal='cosine',
                               noise_std=0.0)
  images, labels = resnet50.synthetic_batch(8, DEV, seed=5)
  return g, model, opt, images, labels


def test_resnet50_structure_and_sparsity(rn50):
  g = rn50[0]
  from rigl_amd.workloads import shapes as WS
  assert [m.name for m in g.get_masks()] == list(WS.resnet50_masks().keys())
  assert sum(m.numel for m in g.get_masks()) == 25502912
  assert sum(m.sum() for m in g.get_masks()) == 5100630        # ERK 0.8 non-zeros


def test_resnet50_every_conv_in_situ_and_loss_vs_cpu_oracle(rn50, monkeypatch):
  """(a) Every one of the 54 masked layers, with the operands it actually sees
  inside the network (real shapes, strides, paddings, bf16 activations), against
  torch's fp32 convolution of the same operands on the GPU.  (b) The loss against
  the fp32 CPU restatement of the reference model.  A whole-network gradient
  comparison is meaningless here: at initialisation a 50-layer BN network with
  batch 8 amplifies bf16 rounding chaotically (fp32 vs bf16-rounded CPU models
  decorrelate to cos 0.6 in the first layers with no kernel involved)."""
  import torch.nn.functional as F
  g, model, opt, images, labels = rn50
  from rigl_amd import ops, pruning_layers as PL
  from oracle.resnet_cpu import ResNet50CPU
  # one autograd node per conv (the three strided projections otherwise share a node with their block's conv1, whose
  # internals -- the compact projection gradient -- are checked in tests/test_conv_pair_gpu.py)
  monkeypatch.setattr(PL, '_CONV_PAIR', False)
  g.refresh_shadows(force=True)
  for v in g.variables.values():
    if v.name.endswith('bn3/gamma:0'):
      v.data.fill_(0.5)
  rec = []
  fwd0, bwd0 = PL._MaskedConvFn.forward, PL._MaskedConvFn.backward

  def fwd(ctx, x, lv, desc, need_dx, want_stats=False, bn_holder=None, pending=None):
    out = fwd0(ctx, x, lv, desc, need_dx, want_stats, bn_holder, pending)   # (pending: the conv fills x itself, so x is read after it)
    y = out[0] if want_stats else out
    rec.append(dict(lv=lv, d=desc, x=x.detach().clone(), y=y.detach().clone()))
    ctx.rec = rec[-1]
    return out

  def bwd(ctx, dy, _dpart=None):
    out = bwd0(ctx, dy)
    ctx.rec.update(dy=dy.detach().clone(), dx=None if out[0] is None else out[0].detach().clone(),
                   dw=ctx.lv.weights.grad.detach().clone())
    return out

  # the convs that share their input with a shortcut (fused gradient accumulation)
  ffwd0, fbwd0 = PL._MaskedConvForkFn.forward, PL._MaskedConvForkFn.backward

  def ffwd(ctx, x, lv, desc, want_stats=False, bn_holder=None):
    out = ffwd0(ctx, x, lv, desc, want_stats, bn_holder)
    rec.append(dict(lv=lv, d=desc, x=x.detach().clone(), y=out[0].detach().clone()))
    ctx.rec = rec[-1]
    return out

  def fbwd(ctx, dy, dalias, _dpart=None):
    out = fbwd0(ctx, dy, dalias)
    ctx.rec.update(dy=dy.detach().clone(), dx=out[0].detach().clone(), dw=ctx.lv.weights.grad.detach().clone(),
                   dadd=None if dalias is None else dalias.detach().clone())
    return out

  PL._MaskedConvFn.forward, PL._MaskedConvFn.backward = staticmethod(fwd), staticmethod(bwd)
  PL._MaskedConvForkFn.forward, PL._MaskedConvForkFn.backward = staticmethod(ffwd), staticmethod(fbwd)
  try:
    loss = model.loss(images, labels, label_smoothing=0.1)
    opt.compute_gradients(loss)
  finally:
    PL._MaskedConvFn.forward, PL._MaskedConvFn.backward = staticmethod(fwd0), staticmethod(bwd0)
    PL._MaskedConvForkFn.forward, PL._MaskedConvForkFn.backward = staticmethod(ffwd0), staticmethod(fbwd0)
  assert sum(1 for r in rec if r.get('dadd') is not None) == 16      # one per bottleneck block
  assert len(rec) == 54
  for r in rec:
    d, lv = r['d'], r['lv']
    w = lv.hwio.float().reshape(d.kh, d.kw, d.cin, d.cout).permute(3, 2, 0, 1).contiguous().requires_grad_(True)
    x = r['x'].float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    pb = max((d.ho - 1) * d.stride_h + d.kh - d.h - d.pad_top, 0)
    pr = max((d.wo - 1) * d.stride_w + d.kw - d.w - d.pad_left, 0)
    y = F.conv2d(F.pad(x, (d.pad_left, pr, d.pad_top, pb)), w, stride=(d.stride_h, d.stride_w))[:, :, :d.ho, :d.wo]
    y.backward(r['dy'].float().permute(0, 3, 1, 2))
    yr = y.detach().permute(0, 2, 3, 1)
    tag = lv.scope
    assert (r['y'].float() - yr).abs().max() <= 2.0**-7 * yr.abs().max() + 1e-6, tag
    dwr = w.grad.permute(2, 3, 1, 0)
    assert (r['dw'].reshape(dwr.shape) - dwr).abs().max() <= 3e-4 * dwr.abs().max() + 1e-9, tag
    if r['dx'] is not None:
      dxr = x.grad.permute(0, 2, 3, 1)
      if r.get('dadd') is not None:
        dxr = dxr + r['dadd'].float()
      assert (r['dx'].float() - dxr).abs().max() <= 2.0**-7 * dxr.abs().max() + 1e-9, tag
  # (b) loss vs the CPU model with the same (bf16-rounded, masked) weights
  cpu = ResNet50CPU(seed=0)
  for i, l in enumerate(g.layers):
    cpu.w[i] = l.hwio.float().cpu().reshape(l.weights.shape if len(l.weights.shape) == 4 else (1, 1) + tuple(l.weights.shape))
    cpu.m[i] = torch.ones_like(cpu.w[i])
  for b in cpu.blocks:
    cpu.bn[b['c3'][1]][0].data.fill_(0.5)
  torch.set_num_threads(min(torch.get_num_threads(), 32))
  with torch.no_grad():
    loss_cpu = F.cross_entropy(cpu.forward(images.float().cpu().permute(0, 3, 1, 2)), labels.cpu(), label_smoothing=0.1)
  assert abs(float(loss.detach()) - float(loss_cpu)) < 2e-2 * abs(float(loss_cpu))


def test_resnet50_optimizer_and_mask_update_bit_exact(rn50):
  """gs=0 is not an update step (begin=1): apply; gs=1 is: prune/regrow.
  Both compared bit-for-bit with the oracle fed the device's own gradients."""
  g, model, opt, images, labels = rn50
  gs = g.get_or_create_global_step()
  gs.value = 0
  opt._last_update_step = -100
  from rigl_amd import variables as V
  # ---- step A: masked Nesterov momentum (K3) over the arenas
  loss = model.loss(images, labels, label_smoothing=0.1)
  gv = opt.compute_gradients(loss)
  inner = opt._optimizer
  inner._ensure_slots()
  W0, G0, A0 = g.W.cpu().numpy(), g.G.cpu().numpy(), inner._slot.cpu().numpy()
  bits = g.BITS.cpu().numpy().view(np.uint32)
  opt.apply_gradients(gv, gs)
  assert gs.value == 1
  b, e = g.seg[V.KIND_MASKED]
  m = np.unpackbits(bits.view(np.uint8), bitorder='little')[:e - b].astype(np.float32)
  w_ref, a_ref = O.momentum_apply(W0[b:e], A0[b:e], O.masked_grad(G0[b:e], m, W0[b:e], 1e-4), 0.05, 0.9)
  np.testing.assert_array_equal(g.W[b:e].cpu().numpy().view(np.uint32), w_ref.view(np.uint32))
  np.testing.assert_array_equal(inner._slot[b:e].cpu().numpy().view(np.uint32), a_ref.view(np.uint32))
  b, e = g.seg[V.KIND_OTHER]
  w_ref, a_ref = O.momentum_apply(W0[b:e], A0[b:e], G0[b:e], 0.05, 0.9)
  np.testing.assert_array_equal(g.W[b:e].cpu().numpy().view(np.uint32), w_ref.view(np.uint32))
  # ---- step B: mask update (K2), all 54 layers in one call
  loss = model.loss(images, labels, label_smoothing=0.1)
  gv = opt.compute_gradients(loss)
  layers = g.masked_layers()
  before = [(l.mask.numpy().copy(), l.weights.numpy().copy(), l.weights.grad.cpu().numpy().copy(),
             inner.get_slot(l.weights, 'momentum').cpu().numpy().copy()) for l in layers]
  opt.apply_gradients(gv, gs)
  assert gs.value == 1                      # update iterations do not advance global_step (F9)
  frac = O.get_drop_fraction('cosine', 0.3, 1, 1, 25000, True)
  assert np.float32(opt.drop_fraction) == np.float32(frac)
  counts = opt.last_counts.cpu().numpy()
  for i, (l, (m0, w0, g0, a0)) in enumerate(zip(layers, before)):
    r = O.rigl_mask_update(m0, w0, g0, frac, momentum=a0)
    np.testing.assert_array_equal(l.mask.numpy(), r['mask'], err_msg=l.scope)
    np.testing.assert_array_equal(l.weights.numpy().view(np.uint32), r['weights'].view(np.uint32), err_msg=l.scope)
    np.testing.assert_array_equal(inner.get_slot(l.weights, 'momentum').cpu().numpy(), r['momentum'])
    assert tuple(counts[i][:3]) == (r['n_ones'], r['n_prune'], r['n_keep'])
  assert sum(m.sum() for m in g.get_masks()) == 5100630        # connections conserved


def test_resnet50_training_reduces_loss(rn50):
  g, model, opt, images, labels = rn50
  gs = g.get_or_create_global_step()
  first = last = None
  opt._optimizer._lr = 0.004            # batch 8: momentum 0.9 makes 0.05 unstable
  for i in range(40):
    loss = model.loss(images, labels, label_smoothing=0.1)
    opt.minimize(loss, gs)
    v = float(loss.detach())
    assert np.isfinite(v)
    first = v if first is None else first
    last = v
  assert last < first - 0.5, (first, last)     # memorising 8 images
