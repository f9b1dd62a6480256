"""Fused BN (+residual) (+ReLU) glue kernels vs PyTorch's batch_norm / relu in
fp32 on the same bf16 operands (forward, backward, running statistics)."""
import pytest

torch = pytest.importorskip('torch')
import torch.nn.functional as F  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _ref(x, res, gamma, beta, relu, dy, eps=1e-5):
  xr = x.float().requires_grad_(True)
  rr = res.float().requires_grad_(True) if res is not None else None
  g = gamma.clone().requires_grad_(True)
  b = beta.clone().requires_grad_(True)
  rm, rv = torch.zeros_like(gamma), torch.ones_like(gamma)
  c = x.shape[-1]
  y = F.batch_norm(xr.reshape(-1, c), rm, rv, g, b, True, 0.1, eps).reshape(x.shape)
  if rr is not None:
    y = y + rr
  if relu:
    y = F.relu(y)
  y.backward(dy.float())
  return y.detach(), xr.grad, (rr.grad if rr is not None else None), g.grad, b.grad, rm, rv


@pytest.mark.parametrize('shape', [(4, 14, 14, 64), (2, 7, 7, 2048), (3, 9, 5, 16), (1, 1, 1, 8), (2, 28, 28, 256),
                                   (33, 3, 3, 40), (2, 56, 56, 64)])
@pytest.mark.parametrize('relu,with_res', [(False, False), (True, False), (True, True), (False, True)])
def test_fused_bn(shape, relu, with_res):
  from rigl_amd import ops
  if shape[0] * shape[1] * shape[2] == 1:
    pytest.skip('batch statistics of one row are degenerate')
  gen = torch.Generator(device=DEV).manual_seed(sum(shape) + relu + 2 * with_res)
  c = shape[-1]
  x = (torch.randn(shape, generator=gen, device=DEV) * 1.7 + 0.3).to(torch.bfloat16)
  res = torch.randn(shape, generator=gen, device=DEV).to(torch.bfloat16) if with_res else None
  dy = torch.randn(shape, generator=gen, device=DEV).to(torch.bfloat16)
  gamma = torch.rand(c, generator=gen, device=DEV) + 0.5
  beta = torch.randn(c, generator=gen, device=DEV) * 0.2
  rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
  y, saved = ops.bn_fwd(x, gamma, beta, rm, rv, 0.1, 1e-5, relu, res)
  dgamma, dbeta = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
  dx, dres = ops.bn_bwd(x, y if (relu and with_res) else None, dy, gamma, saved, relu, dgamma, dbeta,
                        want_dres=with_res)
  yr, dxr, dresr, dgr, dbr, rmr, rvr = _ref(x, res, gamma, beta, relu, dy)

  def close(a, b, rel, what):
    err = (a.float() - b).abs().max().item()
    tol = rel * b.abs().max().item() + 1e-6
    assert err <= tol, '%s: %g > %g' % (what, err, tol)

  close(y, yr, 2.0**-7, 'y')
  # a handful of elements sit within bf16 rounding of the ReLU knee; exclude exact-zero disagreements
  agree = ((y.float() > 0) == (yr > 0)) if relu else torch.ones_like(yr, dtype=torch.bool)
  assert agree.float().mean() > 0.999
  close(dx.float() * agree, dxr * agree, 2e-2, 'dx')
  close(dgamma, dgr, 2e-2, 'dgamma')
  close(dbeta, dbr, 2e-2, 'dbeta')
  if with_res:
    close(dres.float() * agree, dresr * agree, 2.0**-7, 'dres')
  close(rm, rmr, 1e-3, 'running_mean')
  close(rv, rvr, 1e-3, 'running_var')
  close(saved[0], x.float().reshape(-1, c).mean(0), 1e-4, 'saved mean')


@pytest.mark.parametrize('shape', [(4, 14, 14, 64), (3, 9, 5, 16), (2, 28, 28, 256)])
@pytest.mark.parametrize('with_res', [True, False])
def test_fused_bn_relu_bitmask_equals_saved_output(shape, with_res):
  """The 1-bit ReLU mask the forward can leave (1/16 of y's bytes) gives exactly
  the backward that reading y gives."""
  from rigl_amd import ops
  gen = torch.Generator(device=DEV).manual_seed(3 + sum(shape))
  c = shape[-1]
  x = (torch.randn(shape, generator=gen, device=DEV) * 1.7 + 0.3).to(torch.bfloat16)
  res = torch.randn(shape, generator=gen, device=DEV).to(torch.bfloat16) if with_res else None
  dy = torch.randn(shape, generator=gen, device=DEV).to(torch.bfloat16)
  gamma = torch.rand(c, generator=gen, device=DEV) + 0.5
  beta = torch.randn(c, generator=gen, device=DEV) * 0.2
  y, saved, bits = ops.bn_fwd(x, gamma, beta, None, None, 0.1, 1e-5, True, res, want_relu_bits=True)
  y0, saved0 = ops.bn_fwd(x, gamma, beta, None, None, 0.1, 1e-5, True, res)
  assert torch.equal(y, y0) and torch.equal(saved, saved0)
  want = (y.reshape(-1, 8) > 0).to(torch.int32)
  want = (want << torch.arange(8, device=DEV, dtype=torch.int32)).sum(1).to(torch.uint8)
  assert torch.equal(bits, want)
  dg0, db0, dg1, db1 = (torch.empty(c, device=DEV) for _ in range(4))
  dx0, dres0 = ops.bn_bwd(x, y, dy, gamma, saved, True, dg0, db0, want_dres=with_res)
  dx1, dres1 = ops.bn_bwd(x, None, dy, gamma, saved, True, dg1, db1, want_dres=with_res, relu_bits=bits)
  assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
  if with_res:
    assert torch.equal(dres0, dres1)


def test_fused_bn_is_deterministic():
  from rigl_amd import ops
  x = torch.randn(8, 28, 28, 128, device=DEV).to(torch.bfloat16)
  gamma, beta = torch.ones(128, device=DEV), torch.zeros(128, device=DEV)
  y1, s1 = ops.bn_fwd(x, gamma, beta, None, None, 0.1, 1e-5, True)
  y2, s2 = ops.bn_fwd(x, gamma, beta, None, None, 0.1, 1e-5, True)
  assert torch.equal(y1, y2) and torch.equal(s1, s2)


@pytest.mark.parametrize('shape', [(4, 14, 14, 64), (16, 28, 28, 128), (128, 20, 20, 24)])
def test_fused_bn_from_producer_partials(shape):
  """rigl_bn_fwd_stats with the partial sums a conv epilogue would leave (here
  built by hand, one partial per 128 rows: 7 / 98 / 400 of them, covering both
  finalize variants) matches the self-reducing rigl_bn_fwd."""
  from rigl_amd import ops
  gen = torch.Generator(device=DEV).manual_seed(sum(shape))
  c = shape[-1]
  x = (torch.randn(shape, generator=gen, device=DEV) * 1.3 - 0.2).to(torch.bfloat16)
  gamma = torch.rand(c, generator=gen, device=DEV) + 0.5
  beta = torch.randn(c, generator=gen, device=DEV) * 0.2
  xf = x.float().reshape(-1, c)
  m = xf.shape[0]
  parts = (m + 127) // 128
  pad = torch.zeros(parts * 128 - m, c, device=DEV)
  xt = torch.cat([xf, pad]).reshape(parts, 128, c)
  partials = torch.stack([xt.sum(1), (xt * xt).sum(1)], dim=1).contiguous()
  rm0, rv0 = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
  rm1, rv1 = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
  y0, s0 = ops.bn_fwd(x, gamma, beta, rm0, rv0, 0.1, 1e-5, True)
  y1, s1 = ops.bn_fwd(x, gamma, beta, rm1, rv1, 0.1, 1e-5, True, partials=partials)
  assert torch.allclose(s0, s1, rtol=2e-5, atol=1e-6)
  assert torch.allclose(rm0, rm1, rtol=1e-5, atol=1e-7) and torch.allclose(rv0, rv1, rtol=1e-5, atol=1e-7)
  # outputs differ at most by one bf16 rounding where the fp32 value sits on a tie
  assert ((y0.float() - y1.float()).abs() <= 2.0**-7 * y0.float().abs() + 1e-6).all()
  assert (y0 != y1).float().mean() < 1e-3


# ---------------------------------------------------------------- backward reductions riding in the dgrad epilogue
@pytest.mark.parametrize('case', [
    # N, H, W, Cin, Cout, k, stride, relu, residual(-> 1-bit mask), addend
    (4, 14, 14, 256, 256, 3, 1, True, False, False),      # bn1 -> 3x3 conv, mask recomputed from x
    (4, 14, 14, 64, 256, 1, 1, True, False, False),       # narrow 128x64 tile
    (3, 28, 28, 128, 128, 3, 2, True, False, False),      # strided 3x3: parity-class rows
    (3, 28, 28, 256, 512, 1, 2, True, True, True),        # projection: 1x1 / 2 (three all-zero classes), bit mask, addend
    (4, 14, 14, 1024, 256, 1, 1, True, True, True),       # block output feeding conv1 + shortcut: bit mask + addend
    (2, 9, 13, 40, 72, 3, 1, False, False, False),        # ragged rows / columns, no ReLU
    (5, 7, 7, 2048, 512, 1, 1, True, True, False),
])
def test_bn_bwd_reductions_in_the_dgrad_epilogue(case):
  """rigl_masked_conv2d_bwd_bn + rigl_bn_bwd_stats against the unfused pair: the conv's dX and dW
  keep their bits; the batch norm's dgamma / dbeta agree with an fp64 reduction to fp32-accumulation accuracy
  (1e-5 x the sum of the absolute terms) for BOTH paths, and its dx to one bf16 ulp of the unfused result."""
  from rigl_amd import ops
  # the reduction epilogue lives in the igemm dgrad body (opt-in RIGL_BN_FUSE_BWD=1): keep these layers' backward off the
  # ping-pong launch, whose dgrad has none (rigl_conv2d_dgrad_stats_parts returns 0 there and the model does not fuse)
  ops.tune_set('pp_bwd', 0)
  try:
    _bn_bwd_reductions_case(case)
  finally:
    ops.tune_unset('pp_bwd')


def _bn_bwd_reductions_case(case):
  from rigl_amd import ops
  N, H, W, Cin, Cout, k, stride, relu, has_res, has_add = case
  g = torch.Generator(device=DEV).manual_seed(sum(int(v) for v in case))
  pad = (k - 1) // 2
  Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
  d = ops.conv_desc(N, H, W, Cin, Cout, k, k, stride, pad, pad, Ho, Wo)
  x_bn = (torch.randn(N, H, W, Cin, generator=g, device=DEV) * 1.5 + 0.3).to(torch.bfloat16)
  gamma = 1.0 + 0.3 * torch.randn(Cin, generator=g, device=DEV)
  beta = 0.2 * torch.randn(Cin, generator=g, device=DEV)
  res = torch.randn(N, H, W, Cin, generator=g, device=DEV).to(torch.bfloat16) if has_res else None
  rm, rv = torch.zeros(Cin, device=DEV), torch.ones(Cin, device=DEV)
  if has_res:
    y, saved, bits = ops.bn_fwd(x_bn, gamma, beta, rm, rv, 0.1, 1e-5, relu, res, want_relu_bits=True)
  else:
    (y, saved), bits = ops.bn_fwd(x_bn, gamma, beta, rm, rv, 0.1, 1e-5, relu, res), None
  dy = torch.randn(N, Ho, Wo, Cout, generator=g, device=DEV).to(torch.bfloat16)
  w = (torch.randn(k * k * Cin * Cout, generator=g, device=DEV) * (k * k * Cin) ** -0.5)
  hwio = torch.empty_like(w, dtype=torch.bfloat16)
  ohwi = torch.empty_like(hwio)
  ops.pack_weights(w, None, k * k * Cin, Cout, hwio, ohwi)
  addend = torch.randn(N, H, W, Cin, generator=g, device=DEV).to(torch.bfloat16) if has_add else None
  assert ops.dgrad_stats_parts(d) > 0
  dw0, dw1 = torch.empty_like(w), torch.empty_like(w)
  dx0 = ops.conv_bwd(d, y, dy, hwio, dw0, need_dx=True, addend=addend)
  req = dict(x=x_bn, saved=saved, relu=relu, relu_bits=bits)
  dx1 = ops.conv_bwd(d, y, dy, hwio, dw1, need_dx=True, addend=addend, bn_fuse=req)
  assert req.get('partials') is not None and req['partials'].shape == (ops.dgrad_stats_parts(d), 2, Cin)
  assert torch.equal(dx0, dx1) and torch.equal(dw0, dw1)
  dg0, db0, dg1, db1 = [torch.empty(Cin, device=DEV) for _ in range(4)]
  bx0, br0 = ops.bn_bwd(x_bn, None, dx0, gamma, saved, relu, dg0, db0, want_dres=has_res, relu_bits=bits)
  bx1, br1 = ops.bn_bwd(x_bn, None, dx1, gamma, saved, relu, dg1, db1, want_dres=has_res, relu_bits=bits,
                        partials=req['partials'])
  # fp64 reference of the two reductions
  xf, dz = x_bn.double().reshape(-1, Cin), dx0.double().reshape(-1, Cin)
  if relu:
    on = (y.reshape(-1, Cin) > 0)
    dz = torch.where(on, dz, torch.zeros_like(dz))
  xhat = (xf - saved[0].double()) * saved[1].double()
  r0, r1 = dz.sum(0), (dz * xhat).sum(0)
  t0, t1 = dz.abs().sum(0), (dz * xhat).abs().sum(0)
  for db, dg in ((db0, dg0), (db1, dg1)):
    assert ((db.double() - r0).abs() <= 1e-5 * t0 + 1e-6).all()
    assert ((dg.double() - r1).abs() <= 1e-5 * t1 + 1e-6).all()
  a, b = bx1.float(), bx0.float()
  assert ((a - b).abs() <= 2.0**-7 * b.abs() + 1e-6 * b.abs().max()).all()
  assert (bx1 != bx0).float().mean() < 0.02          # and almost every element keeps its bits
  if has_res:
    assert torch.equal(br0, br1)                     # dres = dz does not depend on the reductions


@pytest.mark.parametrize('shape', [(2, 16, 16, 64), (3, 15, 13, 64), (2, 9, 12, 8), (1, 6, 6, 256), (4, 112, 112, 64),
                                   (2, 7, 7, 32)])
def test_bn_relu_maxpool_in_one_piece(shape):
  """relu(bn(x)) -> 3x3 / 2 'SAME' max pooling without the activated tensor (rigl_bn_relu_maxpool_fwd / _bwd, the
  ResNet stem's tail) against the three-kernel form it replaces (bn_fwd, maxpool_fwd; maxpool_bwd, bn_bwd): pooled
  output, argmax bytes, saved statistics and moving averages bit-identical; dx / dgamma / dbeta equal up to the fp32
  summation order of the two reductions (the gathered gradient is rounded to bf16 at the same point)."""
  from rigl_amd import ops
  from rigl_amd.workloads import nn as gnn
  n, h, w, c = shape
  gen = torch.Generator(device=DEV).manual_seed(sum(shape))
  x = (torch.randn(shape, generator=gen, device=DEV) * 1.3 + 0.2).to(torch.bfloat16)
  gamma = torch.rand(c, generator=gen, device=DEV) + 0.5
  beta = torch.randn(c, generator=gen, device=DEV) * 0.3
  d = gnn._pool_desc(x)
  dyp = torch.randn((n, d.ho, d.wo, c), generator=gen, device=DEV).to(torch.bfloat16)
  # three-kernel form
  rm0, rv0 = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
  a, saved0 = ops.bn_fwd(x, gamma, beta, rm0, rv0, 0.1, 1e-5, True)
  y0, arg0 = ops.maxpool_fwd(d, a)
  da = ops.maxpool_bwd(d, dyp, arg0)
  dg0, db0 = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
  dx0, _ = ops.bn_bwd(x, None, da, gamma, saved0, True, dg0, db0)
  # one piece
  rm1, rv1 = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
  y1, arg1, saved1 = ops.bn_relu_maxpool_fwd(d, x, gamma, beta, rm1, rv1, 0.1, 1e-5)
  assert torch.equal(saved0, saved1) and torch.equal(rm0, rm1) and torch.equal(rv0, rv1)
  assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
  assert torch.equal(arg0, arg1)
  dg1, db1 = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
  dx1 = ops.bn_relu_maxpool_bwd(d, x, dyp, arg1, gamma, saved1, dg1, db1)
  scale = lambda t: t.abs().max().item() + 1e-12
  assert (dg1 - dg0).abs().max().item() <= 1e-5 * scale(dg0) + 1e-6
  assert (db1 - db0).abs().max().item() <= 1e-5 * scale(db0) + 1e-6
  # dx: same formula on coefficients that differ in their last fp32 bits -> at most an occasional bf16 ulp
  diff = (dx1.float() - dx0.float()).abs()
  assert (diff <= 2.0**-7 * dx0.float().abs() + 1e-6 * scale(dx0.float())).all()
  assert (diff > 0).float().mean().item() < 0.02
  # deterministic
  dg2, db2 = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
  dx2 = ops.bn_relu_maxpool_bwd(d, x, dyp, arg1, gamma, saved1, dg2, db2)
  assert torch.equal(dx1.view(torch.int16), dx2.view(torch.int16)) and torch.equal(dg1, dg2) and torch.equal(db1, db2)


def test_bn_relu_maxpool_autograd_node_equals_the_two_nodes():
  """nn.bn_relu_max_pool_3x3_s2_same through autograd (fused node vs BatchNorm + max pool nodes)."""
  from rigl_amd import variables as V
  from rigl_amd.workloads import nn as gnn
  outs = []
  for fused in (True, False):
    gnn._STEM_TAIL_FUSED = fused
    g = V.Graph(DEV)
    bn = gnn.BatchNorm(g, 'bn', 64)
    g.finalize()
    gen = torch.Generator(device=DEV).manual_seed(5)
    bn.gamma.data.copy_(torch.rand(64, generator=gen, device=DEV) + 0.5)
    bn.beta.data.copy_(torch.randn(64, generator=gen, device=DEV) * 0.2)
    x = torch.randn((4, 20, 20, 64), generator=gen, device=DEV).to(torch.bfloat16).requires_grad_(True)
    y = gnn.bn_relu_max_pool_3x3_s2_same(bn, x, True)
    dy = torch.randn(y.shape, generator=gen, device=DEV).to(torch.bfloat16)
    y.backward(dy)
    outs.append((y.detach(), x.grad.float(), bn.gamma.grad.clone(), bn.beta.grad.clone(), bn.moving_mean.clone()))
  gnn._STEM_TAIL_FUSED = True
  (y1, dx1, dg1, db1, mm1), (y0, dx0, dg0, db0, mm0) = outs
  assert torch.equal(y1.view(torch.int16), y0.view(torch.int16)) and torch.equal(mm1, mm0)
  assert (dg1 - dg0).abs().max() <= 1e-5 * dg0.abs().max() + 1e-6
  assert (db1 - db0).abs().max() <= 1e-5 * db0.abs().max() + 1e-6
  assert ((dx1 - dx0).abs() <= 2.0**-7 * dx0.abs() + 1e-6 * dx0.abs().max()).all()


@pytest.mark.parametrize('shape', [(4, 14, 14, 64), (2, 7, 7, 2048), (3, 9, 5, 16), (2, 28, 28, 256), (33, 3, 3, 40),
                                   (8, 56, 56, 256)])
def test_two_batch_norms_meeting_in_one_add(shape):
  """relu(bn(x) + bn2(x2)) in one piece (rigl_bn_add_bn_fwd / _bwd, the projection-shortcut blocks) against the calls
  it replaces -- bn_fwd(x2) -> bn_fwd(x, residual) and bn_bwd(want_dres) -> bn_bwd(dres): same arithmetic in the same
  order, so output, ReLU bits, statistics, moving averages, both input gradients and all four parameter gradients are
  compared bit for bit."""
  from rigl_amd import ops
  gen = torch.Generator(device=DEV).manual_seed(sum(shape) + 7)
  c = shape[-1]
  x = (torch.randn(shape, generator=gen, device=DEV) * 1.7 + 0.3).to(torch.bfloat16)
  x2 = (torch.randn(shape, generator=gen, device=DEV) * 0.8 - 0.1).to(torch.bfloat16)
  dy = torch.randn(shape, generator=gen, device=DEV).to(torch.bfloat16)
  gamma, gamma2 = torch.rand(c, generator=gen, device=DEV) + 0.5, torch.rand(c, generator=gen, device=DEV) + 0.5
  beta, beta2 = torch.randn(c, generator=gen, device=DEV) * 0.3, torch.randn(c, generator=gen, device=DEV) * 0.3
  mk = lambda: (torch.zeros(c, device=DEV), torch.ones(c, device=DEV))
  # separate calls
  (rm, rv), (rm2, rv2) = mk(), mk()
  sc, s2 = ops.bn_fwd(x2, gamma2, beta2, rm2, rv2, 0.1, 1e-5, False)
  y0, s1, bits0 = ops.bn_fwd(x, gamma, beta, rm, rv, 0.1, 1e-5, True, sc, want_relu_bits=True)
  g = [torch.empty(c, device=DEV) for _ in range(4)]
  dx0, dres = ops.bn_bwd(x, None, dy, gamma, s1, True, g[0], g[1], want_dres=True, relu_bits=bits0)
  dx20, _ = ops.bn_bwd(x2, None, dres, gamma2, s2, False, g[2], g[3])
  # one piece
  (qm, qv), (qm2, qv2) = mk(), mk()
  t2 = ops.bn_statistics(x2, gamma2, beta2, qm2, qv2, 0.1, 1e-5)
  y1, t1, bits1 = ops.bn_add_bn_fwd(x, x2, t2, gamma, beta, qm, qv, 0.1, 1e-5, True)
  h = [torch.empty(c, device=DEV) for _ in range(4)]
  dx1, dx21 = ops.bn_add_bn_bwd(x, x2, bits1, dy, gamma, t1, gamma2, t2, h[0], h[1], h[2], h[3])
  for a, b in ((s1, t1), (s2, t2), (rm, qm), (rv, qv), (rm2, qm2), (rv2, qv2), (bits0, bits1)):
    assert torch.equal(a, b)
  assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
  assert torch.equal(dx0.view(torch.int16), dx1.view(torch.int16))
  assert torch.equal(dx20.view(torch.int16), dx21.view(torch.int16))
  for a, b in zip(g, h):
    assert torch.equal(a, b)


def test_bottleneck_with_projection_shortcut_fused_equals_unfused():
  """A ResNet-50 bottleneck with a projection shortcut through autograd: the paired batch-norm node vs the two nodes."""
  from rigl_amd import variables as V
  from rigl_amd.workloads import nn as gnn, resnet50 as R, shapes as WS
  res = []
  for fused in (True, False):
    gnn._BN_PAIR_FUSED = fused
    from rigl_amd import pruning_layers as PL
    PL.set_init_seed(0)
    g = V.Graph(DEV)
    convs = [c for grp, n, cs in WS.resnet50_blocks() for c in cs if grp == 2 and n == 0]
    blk = R._Bottleneck(g, convs, 'threshold', 1e-4, 'blk')
    g.finalize()
    g.refresh_shadows()
    gen = torch.Generator(device=DEV).manual_seed(3)
    for bn in (blk.bn1, blk.bn2, blk.bn3, blk.proj_bn):
      bn.gamma.data.copy_(torch.rand(bn.channels, generator=gen, device=DEV) + 0.5)
    x = torch.randn((4, 56, 56, 256), generator=gen, device=DEV).to(torch.bfloat16).requires_grad_(True)
    y = blk(x, True)
    dy = torch.randn(y.shape, generator=gen, device=DEV).to(torch.bfloat16)
    y.backward(dy)
    from rigl_amd import ops
    res.append((y.detach().clone(), x.grad.clone(), g.G.clone()))
  gnn._BN_PAIR_FUSED = True
  (y1, dx1, G1), (y0, dx0, G0) = res
  assert torch.equal(y1.view(torch.int16), y0.view(torch.int16))
  assert torch.equal(dx1.view(torch.int16), dx0.view(torch.int16))
  assert torch.equal(G1, G0)


@pytest.mark.parametrize('shape', [(4, 14, 14, 1024), (2, 7, 7, 2048), (8, 28, 28, 128), (33, 3, 3, 40)])
def test_apply_passes_parameters_in_registers_equal_the_lds_copy(shape):
  """Round 6: the apply passes keep a thread's per-channel parameters in registers where the grid stride is a multiple of
  the channel groups ("bn_regs", default on) instead of staging all C channels' parameters in LDS per workgroup.  Same
  arithmetic on the same values: every output of the forward (+ residual + ReLU + bits), the backward (dx, dres) and the
  two-batch-norm pair must have the bits of the LDS path (resnet_model.py:41-82, 456-501 through autodiff)."""
  from rigl_amd import ops
  gen = torch.Generator(device=DEV).manual_seed(sum(shape))
  c = shape[-1]
  x = (torch.randn(shape, generator=gen, device=DEV) * 1.7 + 0.3).to(torch.bfloat16)
  x2 = torch.randn(shape, generator=gen, device=DEV).to(torch.bfloat16)
  dy = torch.randn(shape, generator=gen, device=DEV).to(torch.bfloat16)
  gamma = torch.rand(c, generator=gen, device=DEV) + 0.5
  beta = torch.randn(c, generator=gen, device=DEV) * 0.2
  outs = []
  try:
    for regs in (1, 0):
      ops.tune_set('bn_regs', regs)
      rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
      y, saved, bits = ops.bn_fwd(x, gamma, beta, rm, rv, 0.1, 1e-5, True, x2, want_relu_bits=True)
      dg, db = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
      dx, dres = ops.bn_bwd(x, None, dy, gamma, saved, True, dg, db, want_dres=True, relu_bits=bits)
      y0, saved0 = ops.bn_fwd(x, gamma, beta, rm.clone(), rv.clone(), 0.1, 1e-5, True, None)
      dx0, _ = ops.bn_bwd(x, None, dy, gamma, saved0, True, dg, db, want_dres=False)
      rm2, rv2 = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
      saved2 = ops.bn_statistics(x2, gamma, beta, rm2, rv2, 0.1, 1e-5)
      yp, savedp, bitsp = ops.bn_add_bn_fwd(x, x2, saved2, gamma, beta, rm.clone(), rv.clone(), 0.1, 1e-5, True)
      dgp, dbp, dgp2, dbp2 = (torch.empty(c, device=DEV) for _ in range(4))
      dxp, dxp2 = ops.bn_add_bn_bwd(x, x2, bitsp, dy, gamma, savedp, gamma, saved2, dgp, dbp, dgp2, dbp2)
      outs.append([t.clone() for t in (y, bits, dx, dres, y0, dx0, yp, bitsp, dxp, dxp2)])
  finally:
    ops.tune_unset('bn_regs')
  for a, b in zip(*outs):
    assert torch.equal(a.view(torch.uint8) if a.dtype != torch.uint8 else a, b.view(torch.uint8) if b.dtype != torch.uint8 else b)
