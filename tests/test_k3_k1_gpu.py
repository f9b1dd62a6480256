"""Parity of K3 (masked SGD-momentum, bit-exact vs the oracle), the weight
shadow packer, and K1 (MFMA masked conv fwd/dgrad/wgrad vs an fp32 / fp64
convolution of the same bf16-rounded operands)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
import torch.nn.functional as F  # noqa: E402

from oracle import rigl_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = 'cuda:0'


def _t(a):
  return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32).reshape(-1)).to(DEV)


def _bf16_round(a):
  return torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).float().numpy()


# ------------------------------------------------------------------ K3
@pytest.mark.parametrize('n', [1, 3, 4, 37, 4096, 1000003])
@pytest.mark.parametrize('nesterov', [True, False])
def test_masked_momentum_bit_exact(n, nesterov):
  from rigl_amd import ops
  rs = np.random.RandomState(n)
  w = rs.randn(n).astype(np.float32)
  a = rs.randn(n).astype(np.float32)
  g = rs.randn(n).astype(np.float32)
  mask = (rs.rand(n) < 0.2).astype(np.float32)
  lr, mu, wd = 0.1, 0.9, 1e-4
  gv = O.masked_grad(g, mask, w, wd)
  w_ref, a_ref = O.momentum_apply(w, a, gv, lr, mu, nesterov=nesterov)
  tw, ta, tg = _t(w), _t(a), _t(g)
  bits = ops.mask_pack(_t(mask))
  shadow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
  ops.masked_sgd_momentum(tw, tg, lr, momentum=ta, mask_bits=bits, mu=mu,
                          weight_decay=wd, nesterov=nesterov, w_shadow=shadow)
  np.testing.assert_array_equal(tw.cpu().numpy().view(np.uint32), w_ref.view(np.uint32))
  np.testing.assert_array_equal(ta.cpu().numpy().view(np.uint32), a_ref.view(np.uint32))
  np.testing.assert_array_equal(shadow.float().cpu().numpy(), _bf16_round(w_ref * mask))


def test_dense_sgd_and_grad_scale():
  from rigl_amd import ops
  rs = np.random.RandomState(1)
  n = 5001
  w, g = rs.randn(n).astype(np.float32), rs.randn(n).astype(np.float32)
  tw = _t(w)
  ops.masked_sgd_momentum(tw, _t(g), 0.05)            # plain GD, no mask
  np.testing.assert_array_equal(tw.cpu().numpy(), O.sgd_apply(w, g, 0.05))
  # DP mean: grad_scale = 1/world applied before the mask/decay
  tw, ta = _t(w), _t(np.zeros(n))
  ops.masked_sgd_momentum(tw, _t(g), 0.05, momentum=ta, mu=0.9, grad_scale=0.125, nesterov=True)
  w_ref, _ = O.momentum_apply(w, np.zeros(n, np.float32), (g * np.float32(0.125)).astype(np.float32), 0.05, 0.9)
  np.testing.assert_array_equal(tw.cpu().numpy(), w_ref)


def test_trajectory_matches_reference_on_gpu():
  """The reference's own toy training run (tests/golden/trajectory.npz,
  produced by executing rigl/sparse_optimizers_base.py) replayed with the HIP
  kernels: masks, weights, momentum bit-identical at every step."""
  import os
  from rigl_amd import ops
  t = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'trajectory.npz'))
  for tag, inner, acc in [('rigl_mom', 'mom', 0.0), ('rigl_mom_acc', 'mom', 0.5), ('rigl_sgd', 'sgd', 0.0)]:
    W, M, GS, FR = t[tag + '__w'], t[tag + '__mask'], t[tag + '__gs'], t[tag + '__frac']
    n_inp, n_out = W.shape[1:]
    tw = _t(W[0])
    bits = ops.mask_pack(_t(M[0]))
    ta = _t(np.zeros_like(W[0])) if inner == 'mom' else None
    sched = O.RigLSchedule(1, 17, 4, 0.4, 'cosine')   # scalar host logic only
    for i in range(len(FR)):
      gs = sched.global_step
      dense = np.broadcast_to((np.arange(n_out, dtype=np.float32) * np.float32(gs)).astype(np.float32),
                              (n_inp, n_out)).astype(np.float32)
      is_upd, frac = sched.step()
      if is_upd:
        ops.prune_regrow([dict(w=tw, momentum=ta, mask_bits=bits, dense_grad=_t(dense))], float(frac),
                         initial_acc_scale=acc)
      else:
        ops.masked_sgd_momentum(tw, _t(dense), 0.01, momentum=ta, mask_bits=bits, mu=0.9, nesterov=True)
      np.testing.assert_array_equal(ops.mask_unpack(bits, (n_inp * n_out,)).cpu().numpy(),
                                    M[i + 1].reshape(-1), err_msg='%s step %d' % (tag, i))
      np.testing.assert_array_equal(tw.cpu().numpy().view(np.uint32), W[i + 1].reshape(-1).view(np.uint32),
                                    err_msg='%s step %d' % (tag, i))
      if inner == 'mom':
        np.testing.assert_array_equal(ta.cpu().numpy(), t[tag + '__mom'][i].reshape(-1))


# ------------------------------------------------------------------ pack
@pytest.mark.parametrize('k,cout', [(64, 64), (147, 64), (9 * 64, 64), (2048, 1000), (100, 10), (1, 8), (4608, 512)])
def test_pack_weights(k, cout):
  from rigl_amd import ops
  rs = np.random.RandomState(k + cout)
  w = rs.randn(k, cout).astype(np.float32)
  mask = (rs.rand(k, cout) < 0.3).astype(np.float32)
  tw = _t(w)
  bits = ops.mask_pack(_t(mask))
  hwio = torch.empty(k * cout, dtype=torch.bfloat16, device=DEV)
  ohwi = torch.empty(k * cout, dtype=torch.bfloat16, device=DEV)
  ops.pack_weights(tw, bits, k, cout, hwio, ohwi)
  ref = _bf16_round(w * mask)
  np.testing.assert_array_equal(hwio.float().cpu().numpy().reshape(k, cout), ref)
  np.testing.assert_array_equal(ohwi.float().cpu().numpy().reshape(cout, k), ref.T)
  ops.pack_weights(tw, None, k, cout, hwio, None)       # no mask = dense
  np.testing.assert_array_equal(hwio.float().cpu().numpy().reshape(k, cout), _bf16_round(w))


# ------------------------------------------------------------------ K1
def _ref_conv(x, w_hwio, dy, stride, pt, pl, ho, wo, dtype=torch.float64, dev='cpu'):
  """fp64 (or fp32 on GPU) convolution of the bf16-rounded operands.
  x [N,H,W,C], w [kh,kw,ci,co], dy [N,Ho,Wo,Co] -> y, dx, dw (same layouts)."""
  N, H, W, C = x.shape
  kh, kw = w_hwio.shape[:2]
  pb = max((ho - 1) * stride + kh - H - pt, 0)
  pr = max((wo - 1) * stride + kw - W - pl, 0)
  xr = x.to(dev, dtype).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
  wr = w_hwio.to(dev, dtype).permute(3, 2, 0, 1).contiguous().requires_grad_(True)
  xp = F.pad(xr, (pl, pr, pt, pb))
  y = F.conv2d(xp, wr, stride=stride)[:, :, :ho, :wo]
  y.backward(dy.to(dev, dtype).permute(0, 3, 1, 2))
  return (y.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1), wr.grad.permute(2, 3, 1, 0))


def _check(name, got, ref, scale_ref, rel):
  """|got - ref| <= rel * scale, scale = sum |a||b| per output (error model of
  an fp32-accumulated dot product) or the bf16 output rounding."""
  got = got.double().cpu()
  ref = ref.double().cpu()
  err = (got - ref).abs()
  tol = rel * scale_ref.double().cpu() + 1e-30
  bad = err > tol
  if bad.any():
    idx = bad.nonzero()[:6]
    raise AssertionError('%s: %d/%d outside tolerance; worst ratio %.3g; first idx %s got %s ref %s' % (
        name, int(bad.sum()), bad.numel(), float((err / tol).max()), idx.tolist(),
        got[tuple(idx.T)].tolist(), ref[tuple(idx.T)].tolist()))


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad_top, pad_left, Ho, Wo
    (2, 14, 14, 64, 64, 1, 1, 0, 0, 14, 14),        # 1x1
    (2, 14, 14, 64, 256, 1, 1, 0, 0, 14, 14),       # 1x1 expand
    (2, 14, 14, 256, 64, 1, 1, 0, 0, 14, 14),       # 1x1 reduce
    (3, 14, 14, 256, 512, 1, 2, 0, 0, 7, 7),        # projection shortcut, stride 2 (fixed_padding k=1 -> none)
    (2, 14, 14, 64, 64, 3, 1, 1, 1, 14, 14),        # 3x3 SAME
    (2, 14, 14, 128, 128, 3, 2, 1, 1, 7, 7),        # 3x3 stride 2, fixed_padding (1,1) + VALID
    (2, 16, 16, 32, 32, 3, 2, 0, 0, 8, 8),          # TF SAME stride 2: pads (0,1) -- WRN
    (2, 9, 11, 16, 48, 3, 1, 1, 1, 9, 11),          # odd sizes, small channels (BK=16)
    (2, 32, 32, 3, 64, 7, 2, 3, 3, 16, 16),         # 7x7/2 stem, Cin=3 -> im2col path
    (2, 32, 32, 3, 16, 3, 1, 1, 1, 32, 32),         # WRN stem
    (5, 1, 1, 2048, 1000, 1, 1, 0, 0, 1, 1),        # final_dense as 1x1 conv
    (1, 7, 7, 512, 512, 3, 1, 1, 1, 7, 7),          # late 3x3
    (2, 10, 10, 40, 72, 3, 1, 1, 1, 10, 10),        # channels % 8 == 0 but not % 16/32/64
    (130, 4, 4, 64, 64, 3, 1, 1, 1, 4, 4),          # many images, M not a tile multiple
    (2, 15, 13, 16, 32, 3, 2, 1, 1, 8, 7),          # odd spatial sizes, stride 2 (parity classes of unequal size)
    (3, 20, 20, 24, 40, 1, 2, 0, 0, 10, 10),        # 1x1 stride 2: three of four parity classes get no tap
    (2, 17, 19, 3, 32, 3, 2, 0, 0, 9, 10),          # tiny-Cin, TF SAME stride 2 on odd sizes
    (2, 12, 12, 4, 16, 5, 1, 2, 2, 12, 12),         # tiny-Cin = 4, 5x5
    (2, 10, 10, 15, 24, 3, 1, 1, 1, 10, 10),        # Cin % 8 != 0 and > 4: explicit im2col path
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case):
  from rigl_amd import ops
  N, H, W, Cin, Cout, k, stride, pt, pl, Ho, Wo = case
  g = torch.Generator().manual_seed(sum(case))
  x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16)
  dy = torch.randn(N, Ho, Wo, Cout, generator=g).to(torch.bfloat16)
  w = torch.randn(k, k, Cin, Cout, generator=g) * (2.0 / (k * k * Cin)) ** 0.5
  mask = (torch.rand(k, k, Cin, Cout, generator=g) < 0.3).float()
  d = ops.conv_desc(N, H, W, Cin, Cout, k, k, stride, pt, pl, Ho, Wo)
  n = k * k * Cin * Cout
  tw = w.reshape(-1).contiguous().to(DEV)
  bits = ops.mask_pack(mask.reshape(-1).contiguous().to(DEV))
  hwio = torch.empty(n, dtype=torch.bfloat16, device=DEV)
  ohwi = torch.empty(n, dtype=torch.bfloat16, device=DEV)
  ops.pack_weights(tw, bits, k * k * Cin, Cout, hwio, ohwi)
  wm = hwio.float().cpu().reshape(k, k, Cin, Cout)          # bf16(mask*w), the operand actually used
  yr, dxr, dwr = _ref_conv(x.float(), wm, dy.float(), stride, pt, pl, Ho, Wo)
  ya, dxa, dwa = _ref_conv(x.float().abs(), wm.abs(), dy.float().abs(), stride, pt, pl, Ho, Wo)
  xd, dyd = x.to(DEV), dy.to(DEV)
  # fwd / dgrad: bf16 outputs -> half-ulp of bf16 on the result + fp32 accumulation noise
  y = ops.conv_fwd(d, xd, ohwi)
  _check('fwd', y.float(), yr, yr.abs() * 2.0**-8 + ya * 1e-5, 1.0)
  if Cin % 8 == 0:
    dx = ops.conv_dgrad(d, dyd, hwio)
    _check('dgrad', dx.float(), dxr, dxr.abs() * 2.0**-8 + dxa * 1e-5, 1.0)
  # wgrad: fp32 output, tolerance 1e-5 relative to sum|x||dy| (north-star: 1e-5 fp32)
  dw = ops.conv_wgrad(d, xd, dyd).reshape(k, k, Cin, Cout)
  _check('wgrad', dw, dwr, dwa, 1e-5)
  # the direct kernels must agree too (independent on-device cross-check)
  y2 = ops.conv_fwd(d, xd, ohwi, force_ref=True)
  _check('fwd_ref', y2.float(), yr, yr.abs() * 2.0**-8 + ya * 1e-5, 1.0)
  if N * H * W * Cin <= 200000:
    dx2 = ops.conv_dgrad(d, dyd, hwio, force_ref=True)
    _check('dgrad_ref', dx2.float(), dxr, dxr.abs() * 2.0**-8 + dxa * 1e-5, 1.0)
    dw2 = ops.conv_wgrad(d, xd, dyd, force_ref=True).reshape(k, k, Cin, Cout)
    _check('wgrad_ref', dw2, dwr, dwa, 1e-5)


@pytest.mark.parametrize('case', [CONV_CASES[0], CONV_CASES[3], CONV_CASES[4], CONV_CASES[5], CONV_CASES[13],
                                  CONV_CASES[14], CONV_CASES[15]])
def test_conv_dgrad_accumulate_is_dgrad_plus_addend(case):
  """rigl_masked_conv2d_dgrad_acc == dgrad followed by a bf16 add, bit for bit
  (stride-1, parity-class stride-2 incl. classes without taps, ragged M)."""
  from rigl_amd import ops
  N, H, W, Cin, Cout, k, stride, pt, pl, Ho, Wo = case
  g = torch.Generator().manual_seed(7 + sum(case))
  dy = torch.randn(N, Ho, Wo, Cout, generator=g).to(torch.bfloat16).to(DEV)
  add = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16).to(DEV)
  hwio = (torch.randn(k * k * Cin * Cout, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
  d = ops.conv_desc(N, H, W, Cin, Cout, k, k, stride, pt, pl, Ho, Wo)
  plain = ops.conv_dgrad(d, dy, hwio)
  fused = ops.conv_dgrad(d, dy, hwio, addend=add)
  want = plain + add
  assert torch.equal(fused.view(torch.int16), want.view(torch.int16))
  # in place on the addend buffer is allowed
  buf = add.clone()
  ops.conv_dgrad(d, dy, hwio, dx=buf, addend=buf)
  assert torch.equal(buf.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize('case', [CONV_CASES[0], CONV_CASES[1], CONV_CASES[4], CONV_CASES[5], CONV_CASES[8],
                                  CONV_CASES[12], CONV_CASES[13], CONV_CASES[18],
                                  (3, 28, 28, 64, 256, 1, 1, 0, 0, 28, 28), (9, 12, 12, 32, 64, 3, 1, 1, 1, 12, 12)])
def test_conv_fwd_epilogue_bn_statistics(case):
  """rigl_masked_conv2d_fwd_stats: y unchanged, and the per-tile partial sums
  add up to the column sums / sums of squares of the bf16 outputs."""
  from rigl_amd import ops
  N, H, W, Cin, Cout, k, stride, pt, pl, Ho, Wo = case
  g = torch.Generator().manual_seed(11 + sum(case))
  x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16).to(DEV)
  ohwi = (torch.randn(k * k * Cin * Cout, generator=g) * (2.0 / (k * k * Cin)) ** 0.5).to(torch.bfloat16).to(DEV)
  d = ops.conv_desc(N, H, W, Cin, Cout, k, k, stride, pt, pl, Ho, Wo)
  y0 = ops.conv_fwd(d, x, ohwi)
  y, part = ops.conv_fwd(d, x, ohwi, stats=True)
  assert torch.equal(y.view(torch.int16), y0.view(torch.int16))
  M = N * Ho * Wo
  assert part.shape == (d._stats_parts, 2, Cout)
  rows = 128                                               # igemm forward: one partial per 128-row tile
  if part.shape[0] != (M + 127) // 128:                    # (kernels with their own tiling leave other row sets: sums only)
    return
  yf = y.double().reshape(M, Cout)
  s = part.double().sum(0)
  np.testing.assert_allclose(s[0].cpu().numpy(), yf.sum(0).cpu().numpy(), rtol=0, atol=2e-6 * float(yf.abs().sum(0).max()) + 1e-6)
  np.testing.assert_allclose(s[1].cpu().numpy(), (yf * yf).sum(0).cpu().numpy(), rtol=2e-6, atol=1e-6)
  # every tile's partial is exactly the fp32 statistics of its own 128 rows (within fp32 summation error)
  t = min(part.shape[0] - 1, 1)
  blk = yf[t * rows:(t + 1) * rows]
  np.testing.assert_allclose(part[t, 0].double().cpu().numpy(), blk.sum(0).cpu().numpy(), rtol=0,
                             atol=1e-5 * float(blk.abs().sum(0).max()) + 1e-6)


@pytest.mark.parametrize('case', [(127, 28, 28, 64, 128, 1, 1, 0, 0, 28, 28),     # 779 tiles of 128x128, ragged last tile
                                  (32, 56, 56, 32, 128, 3, 1, 1, 1, 56, 56)])     # 784 tiles, 3x3 with halo
def test_conv_fwd_256x128_tile_is_bit_identical(case):
  """Forward layers whose 128x128 grid needs a second round of workgroups but whose 256x128 grid does not
  run the 8-wave 256x128 kernel (RIGL_CONV_BIG default rule).  Same K order per output and the same
  128-row statistics partials, so outputs and partials must equal the 128x128 kernel's bit for bit --
  checked in a subprocess with RIGL_CONV_BIG=0, the plan being cached per process."""
  import json
  import subprocess
  import sys
  N, H, W, Cin, Cout, k, stride, pt, pl, Ho, Wo = case
  prog = ("import sys, json, hashlib, torch; sys.path.insert(0, %r); from rigl_amd import ops;"
          "g = torch.Generator().manual_seed(5);"
          "x = torch.randn(%d, %d, %d, %d, generator=g).to(torch.bfloat16).cuda();"
          "w = (torch.randn(%d, generator=g) * %r).to(torch.bfloat16).cuda();"
          "d = ops.conv_desc(%d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d, %d);"
          "y, part = ops.conv_fwd(d, x, w, stats=True); torch.cuda.synchronize();"
          "h = lambda t: hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest();"
          "print(json.dumps({'y': h(y.view(torch.int16)), 'part': h(part.view(torch.int32)), 'shape': list(part.shape),"
          " 'absmax': float(y.float().abs().max())}))"
          % (ROOT, N, H, W, Cin, k * k * Cin * Cout, (2.0 / (k * k * Cin)) ** 0.5,
             N, H, W, Cin, Cout, k, k, stride, pt, pl, Ho, Wo))
  outs = {}
  for big in ('0', '1', '2'):                    # never / the default rule / wherever legal
    env = dict(os.environ, RIGL_CONV_BIG=big, RIGL_ROWSTREAM='0')   # (the 1x1 case is a rowstream.hpp shape by default: this test pins the igemm tiles)
    r = subprocess.run([sys.executable, '-c', prog], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    outs[big] = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
  assert outs['0'] == outs['1'] == outs['2']
  assert outs['1']['shape'] == [(N * Ho * Wo + 127) // 128, 2, Cout] and outs['1']['absmax'] > 0


def test_few_slab_reduce_kernels():
  """Layers with few split-K slabs (large dW, few pixels: the 7x7 3x3 convs) reduce with 4 or 8 split-groups per
  workgroup instead of 16; with groups >= splits every variant is the plain sum in split order.  The standalone plan of
  this shape has 5 slabs (8-group kernel), the shared-launch plan 3 (4-group kernel): both agree with the fp32
  reassociation bound and are run-to-run deterministic."""
  from rigl_amd import ops
  g = torch.Generator().manual_seed(9)
  x = torch.randn(32, 7, 7, 512, generator=g).to(torch.bfloat16).to(DEV)
  dy = torch.randn(32, 7, 7, 512, generator=g).to(torch.bfloat16).to(DEV)
  w = (torch.randn(9 * 512 * 512, generator=g) * 0.02).to(torch.bfloat16).to(DEV)
  d = ops.conv_desc(32, 7, 7, 512, 512, 3, 3, 1, 1, 1, 7, 7)
  dw0 = ops.conv_wgrad(d, x, dy)
  dw1, dw2 = torch.empty_like(dw0), torch.empty_like(dw0)
  ops.conv_bwd(d, x, dy, w, dw1, need_dx=True)
  ops.conv_bwd(d, x, dy, w, dw2, need_dx=True)
  assert torch.equal(dw0.view(torch.int32), ops.conv_wgrad(d, x, dy).view(torch.int32))
  assert torch.equal(dw1.view(torch.int32), dw2.view(torch.int32))
  assert float(dw0.abs().max()) > 0 and float((dw0 - dw1).abs().max()) <= 1e-5 * 32 * 49


@pytest.mark.parametrize('case', [CONV_CASES[1], CONV_CASES[4], CONV_CASES[5], CONV_CASES[8], CONV_CASES[15]])
def test_conv_bwd_single_call_equals_wgrad_then_dgrad(case):
  """rigl_masked_conv2d_bwd == rigl_masked_conv2d_wgrad + rigl_masked_conv2d_dgrad_acc: dX bit for bit, dW to fp32
  reassociation (the fused launch may split the pixel sum over fewer ranges) and deterministic."""
  from rigl_amd import ops
  N, H, W, Cin, Cout, k, stride, pt, pl, Ho, Wo = case
  g = torch.Generator().manual_seed(23 + sum(case))
  x = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16).to(DEV)
  dy = torch.randn(N, Ho, Wo, Cout, generator=g).to(torch.bfloat16).to(DEV)
  hwio = (torch.randn(k * k * Cin * Cout, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
  d = ops.conv_desc(N, H, W, Cin, Cout, k, k, stride, pt, pl, Ho, Wo)
  dw0 = ops.conv_wgrad(d, x, dy)
  dw1 = torch.empty_like(dw0)
  if Cin % 8 == 0:
    add = torch.randn(N, H, W, Cin, generator=g).to(torch.bfloat16).to(DEV)
    dx0 = ops.conv_dgrad(d, dy, hwio, addend=add)
    dx1 = ops.conv_bwd(d, x, dy, hwio, dw1, need_dx=True, addend=add)
    assert torch.equal(dx0.view(torch.int16), dx1.view(torch.int16))
  else:
    assert ops.conv_bwd(d, x, dy, hwio, dw1, need_dx=False) is None
  # same sum, possibly split over fewer pixel ranges than the standalone kernel's plan: fp32 reassociation only
  tol = 1e-5 * float(x.float().abs().mean() * dy.float().abs().mean()) * N * Ho * Wo
  assert float((dw0 - dw1).abs().max()) <= tol
  # on_dw_ready fires once dW's last kernel is enqueued (the data-parallel buckets launch from it)
  if Cin % 8 == 0:
    done = []
    dw2 = torch.full_like(dw0, float('nan'))
    ops.conv_bwd(d, x, dy, hwio, dw2, need_dx=True, on_dw_ready=lambda: done.append(1))
    assert done == [1]
    assert torch.equal(dw2.view(torch.int32), dw1.view(torch.int32))      # deterministic


def test_conv_asymmetric_b_detects_transposes():
  """A = identity-like activations, asymmetric weights: catches swapped
  rows/cols in the MFMA C/D mapping (guide rule 16)."""
  from rigl_amd import ops
  N, H, W, Cin, Cout = 1, 8, 16, 128, 128
  x = torch.zeros(N, H, W, Cin)
  for p in range(H * W):
    x.view(-1, Cin)[p, p % Cin] = 1.0
  w = (torch.arange(Cin).view(Cin, 1) * 3 + torch.arange(Cout).view(1, Cout) * 7).float() % 61 - 30
  d = ops.conv_desc(N, H, W, Cin, Cout, 1, 1, 1, 0, 0, H, W)
  ohwi = w.t().contiguous().to(torch.bfloat16).to(DEV).reshape(-1)
  y = ops.conv_fwd(d, x.to(torch.bfloat16).to(DEV), ohwi).float().cpu().view(-1, Cout)
  ref = x.view(-1, Cin) @ w
  assert torch.equal(y, ref)


@pytest.mark.parametrize('shape', [
    (32, 56, 56, 64, 64, 3, 1, 1, 1, 56, 56),
    (32, 56, 56, 256, 128, 1, 1, 0, 0, 56, 56),
    (32, 28, 28, 512, 1024, 1, 2, 0, 0, 14, 14),
    (32, 14, 14, 256, 256, 3, 1, 1, 1, 14, 14),
    (32, 224, 224, 3, 64, 7, 2, 3, 3, 112, 112),
])
def test_conv_resnet50_sizes_vs_gpu_fp32(shape):
  """ResNet-50 layer shapes at batch 32 against torch's fp32 convolution on the
  same GPU (independent implementation), same bf16-rounded operands."""
  from rigl_amd import ops
  N, H, W, Cin, Cout, k, stride, pt, pl, Ho, Wo = shape
  g = torch.Generator(device=DEV).manual_seed(1)
  x = torch.randn(N, H, W, Cin, generator=g, device=DEV).to(torch.bfloat16)
  dy = torch.randn(N, Ho, Wo, Cout, generator=g, device=DEV).to(torch.bfloat16)
  w = (torch.randn(k, k, Cin, Cout, generator=g, device=DEV) * (2.0 / (k * k * Cin)) ** 0.5)
  d = ops.conv_desc(N, H, W, Cin, Cout, k, k, stride, pt, pl, Ho, Wo)
  n = w.numel()
  hwio = torch.empty(n, dtype=torch.bfloat16, device=DEV)
  ohwi = torch.empty(n, dtype=torch.bfloat16, device=DEV)
  ops.pack_weights(w.reshape(-1).contiguous(), None, k * k * Cin, Cout, hwio, ohwi)
  wm = hwio.float().reshape(k, k, Cin, Cout)
  yr, dxr, dwr = _ref_conv(x.float(), wm, dy.float(), stride, pt, pl, Ho, Wo, dtype=torch.float32, dev=DEV)
  y = ops.conv_fwd(d, x, ohwi).float()
  assert (y - yr).abs().max() <= 2.0**-7 * yr.abs().max()
  dw = ops.conv_wgrad(d, x, dy).reshape(k, k, Cin, Cout)
  assert (dw - dwr).abs().max() <= 2e-4 * dwr.abs().max()      # both sides are fp32 accumulations
  if Cin % 8 == 0:
    dx = ops.conv_dgrad(d, dy, hwio).float()
    assert (dx - dxr).abs().max() <= 2.0**-7 * dxr.abs().max()


# ------------------------------------------------------------------ K1d depthwise
@pytest.mark.parametrize('case', [(2, 14, 14, 32, 3, 1), (2, 15, 13, 64, 3, 2), (3, 8, 8, 8, 3, 2), (1, 7, 7, 1024, 3, 1),
                                  (2, 112, 112, 32, 3, 1), (2, 9, 9, 40, 5, 1),
                                  # MobileNet-v1's own layers (mobilenetv1_model.py:282-298) at a small batch, ragged strips (W = 7, 14)
                                  (3, 112, 112, 64, 3, 2), (2, 56, 56, 128, 3, 1), (2, 56, 56, 128, 3, 2), (2, 28, 28, 256, 3, 2),
                                  (4, 14, 14, 512, 3, 1), (4, 14, 14, 512, 3, 2), (5, 7, 7, 1024, 3, 1),
                                  # TF 'SAME' at stride 2 on an even map: pad (0, 1) -- the even-pad_left branch of the strided dgrad
                                  (2, 16, 12, 48, 3, 2, 'same'), (2, 14, 14, 16, 3, 2, 'same'), (1, 6, 10, 8, 3, 1), (2, 5, 3, 24, 3, 2)])
def test_depthwise_conv(case):
  """Dense depthwise conv fwd/dgrad/wgrad vs a grouped fp32 convolution of the
  same bf16 activations (stride 1 = SAME, stride 2 = fixed_padding + VALID, or TF SAME when the case says so)."""
  from rigl_amd import ops
  N, H, W, C, k, stride = case[:6]
  same = len(case) > 6
  g = torch.Generator().manual_seed(sum(case[:6]))
  x = torch.randn(N, H, W, C, generator=g).to(torch.bfloat16)
  w = torch.randn(k, k, C, 1, generator=g) * 0.3
  pad = (k - 1) // 2
  xr = x.float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
  wr = w.permute(2, 3, 0, 1).contiguous().requires_grad_(True)          # [C,1,k,k]
  if same:                                                              # out = ceil(in / s), extra padding at the end
    Ho, Wo = -(-H // stride), -(-W // stride)
    th, tw = max((Ho - 1) * stride + k - H, 0), max((Wo - 1) * stride + k - W, 0)
    pt, pl = th // 2, tw // 2
    yr = F.conv2d(F.pad(xr, (pl, tw - pl, pt, th - pt)), wr, stride=stride, groups=C)
  else:
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    pt = pl = pad
    yr = F.conv2d(xr, wr, stride=stride, padding=pad, groups=C)
  dy = torch.randn(N, Ho, Wo, C, generator=g).to(torch.bfloat16)
  d = ops.conv_desc(N, H, W, C, C, k, k, stride, pt, pl, Ho, Wo)
  assert yr.shape[2:] == (Ho, Wo)
  yr.backward(dy.float().permute(0, 3, 1, 2))
  wd = w.reshape(-1).contiguous().to(DEV)
  y = ops.depthwise_fwd(d, x.to(DEV), wd).float().cpu()
  dx = ops.depthwise_dgrad(d, dy.to(DEV), wd).float().cpu()
  dw = torch.empty(k * k * C, device=DEV)
  ops.depthwise_wgrad(d, x.to(DEV), dy.to(DEV), dw)
  ref_y, ref_dx = yr.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1)
  ref_dw = wr.grad.permute(2, 3, 0, 1).reshape(-1)
  # per element: bf16 rounding of the output + fp32 accumulation noise relative to the size of the dot product
  assert ((y - ref_y).abs() <= 2.0**-8 * ref_y.abs() + 1e-5 * ref_y.abs().max() + 1e-6).all()
  assert ((dx - ref_dx).abs() <= 2.0**-8 * ref_dx.abs() + 1e-5 * ref_dx.abs().max() + 1e-6).all()
  assert (dw.cpu() - ref_dw).abs().max() <= 1e-4 * ref_dw.abs().max() + 1e-5
  # deterministic: the split reduction of the weight gradient has a fixed order
  dw2 = torch.empty_like(dw)
  ops.depthwise_wgrad(d, x.to(DEV), dy.to(DEV), dw2)
  assert torch.equal(dw, dw2)
  # the forward with the batch-norm statistics epilogue: same y bit for bit, partial rows that add up to the sums of the
  # bf16-rounded outputs, and the same partial rows on a second launch
  ys, part = ops.depthwise_fwd(d, x.to(DEV), wd, stats=True)
  assert torch.equal(ys.float().cpu(), y)
  if k == 3 and C % 8 == 0 and 256 % (C // 8) == 0:
    assert part is not None and part.shape[1:] == (2, C)
    y64 = ys.double().reshape(-1, C)
    tot = part.double().sum(0)
    assert (tot[0] - y64.sum(0)).abs().max() <= 1e-5 * y64.abs().sum(0).max() + 1e-6
    assert (tot[1] - (y64 * y64).sum(0)).abs().max() <= 1e-5 * (y64 * y64).sum(0).max() + 1e-6
    _, part2 = ops.depthwise_fwd(d, x.to(DEV), wd, stats=True)
    assert torch.equal(part, part2)
  else:
    assert part is None


def test_plan_caches_follow_the_knobs_on_a_live_descriptor():
  """the statistics-part count cached on a descriptor follows a knob set AFTER the layer's first call --
  a kernel switched on later writes fewer partial rows than the stale count, and batch norm would have summed
  uninitialised rows.  Same descriptor, rowstream off -> on -> off: the partial tensor's shape follows and the sums hold."""
  from rigl_amd import ops
  g = torch.Generator().manual_seed(4)
  N, H, Ci, Co = 8, 28, 128, 512
  x = torch.randn(N, H, H, Ci, generator=g).to(torch.bfloat16).to(DEV)
  w = (torch.randn(Ci * Co, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
  d = ops.conv_desc(N, H, H, Ci, Co, 1, 1, 1, 0, 0, H, H)
  M = N * H * H
  shapes = []
  try:
    for v in (0, 2, 0):            # (2: every legal layer -- the default rule only takes the row counts it was measured on)
      ops.tune_set('rowstream', v)
      y, part = ops.conv_fwd(d, x, w, stats=True)
      shapes.append(part.shape[0])
      yf = y.double().reshape(M, Co)
      s = part.double().sum(0)
      assert float((s[0] - yf.sum(0)).abs().max()) <= 2e-6 * float(yf.abs().sum(0).max()) + 1e-6
      q = (yf * yf).sum(0)
      assert float(((s[1] - q).abs() / (q + 1e-6)).max()) <= 2e-6
  finally:
    ops.tune_unset('rowstream')
  assert shapes[0] == shapes[2] == (M + 127) // 128 and shapes[1] != shapes[0] and shapes[1] % 8 == 0
