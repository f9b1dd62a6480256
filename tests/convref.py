"""fp64 convolution reference for the K1 parity tests -- TEST INFRASTRUCTURE, never imported by the product.

The convolution is written out as one matrix product per filter tap on padded / strided slices (``dgemm`` on
whatever device the tensors live on), differentiated by autograd, so it shares nothing with the kernels under
test nor with MIOpen (which has no fp64 convolution).  Operands are the SAME bf16-rounded values the kernels
read; the error model of a bf16-in / fp32-accumulate dot product is

    |got - ref| <= 2^-8 |ref|  (bf16 output rounding, fwd / dgrad only)  +  rel * sum_k |a_k| |b_k|

with ``sum |a||b|`` obtained by running the same reference on the absolute values (rigl test convention:
rel = 1e-5, the north star's fp32 tolerance; tests/test_k3_k1_gpu.py:_check).
"""
import torch
import torch.nn.functional as F


def conv_fp64(x, w_hwio, dy, stride, pt, pl, ho, wo, want=('y', 'dx', 'dw')):
  """x [N,H,W,Ci], w [kh,kw,Ci,Co], dy [N,Ho,Wo,Co] (any float dtype, any device) -> dict of fp64 tensors
  y [N,Ho,Wo,Co], dx [N,H,W,Ci], dw [kh,kw,Ci,Co]."""
  N, H, W, Ci = x.shape
  kh, kw, _, Co = w_hwio.shape
  sh, sw = (stride, stride) if isinstance(stride, int) else stride
  pb = max((ho - 1) * sh + kh - H - pt, 0)
  pr = max((wo - 1) * sw + kw - W - pl, 0)
  need_grad = ('dx' in want) or ('dw' in want)
  xd = x.double().detach().requires_grad_(need_grad and 'dx' in want)
  wd = w_hwio.double().detach().requires_grad_(need_grad and 'dw' in want)
  xp = F.pad(xd, (0, 0, pl, pr, pt, pb))                 # pads W then H of an NHWC tensor
  y = None
  for r in range(kh):
    for s in range(kw):
      sl = xp[:, r:r + (ho - 1) * sh + 1:sh, s:s + (wo - 1) * sw + 1:sw, :]
      t = sl.reshape(-1, Ci) @ wd[r, s]
      y = t if y is None else y + t
  y = y.reshape(N, ho, wo, Co)
  out = {'y': y.detach()}
  if need_grad:
    y.backward(dy.double())
    if 'dx' in want:
      out['dx'] = xd.grad
    if 'dw' in want:
      out['dw'] = wd.grad
  return out


def check_close(name, got, ref, absref, rel=1e-5, out_ulp=0.0):
  """|got - ref| <= out_ulp * |ref| + rel * absref  elementwise; raises with the worst offenders."""
  got, ref, absref = got.double(), ref.double(), absref.double()
  err = (got - ref).abs()
  tol = out_ulp * ref.abs() + rel * absref + 1e-30
  bad = err > tol
  if bool(bad.any()):
    idx = bad.nonzero()[:6]
    raise AssertionError('%s: %d/%d outside tolerance; worst ratio %.3g; first idx %s got %s ref %s' % (
        name, int(bad.sum()), bad.numel(), float((err / tol).max()), idx.tolist(),
        got[tuple(idx.T)].tolist(), ref[tuple(idx.T)].tolist()))
  return float((err / tol).max())
