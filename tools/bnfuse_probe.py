#!/usr/bin/env python3
"""Times conv_bwd (one launch: dgrad + wgrad) with and without the batch-norm reductions in the dgrad epilogue, and the
stand-alone reduction kernel it replaces.  Development tool."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import ops  # noqa: E402


def timeit(fn, iters=30, warmup=5):
  for _ in range(warmup):
    fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) / iters * 1e3


def main():
  dev = 'cuda:0'
  N = 128
  for name, (H, W, Cin, Cout, k, s) in dict(g3c2=(14, 14, 256, 256, 3, 1), g3c3=(14, 14, 256, 1024, 1, 1), g4c2=(7, 7, 512, 512, 3, 1),
                                            g2c2=(28, 28, 128, 128, 3, 1), g1c2=(56, 56, 64, 64, 3, 1), g1c1=(56, 56, 256, 64, 1, 1)).items():
    p = (k - 1) // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    d = ops.conv_desc(N, H, W, Cin, Cout, k, k, s, p, p, Ho, Wo)
    x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
    xbn = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(N, Ho, Wo, Cout, device=dev).to(torch.bfloat16)
    w = torch.randn(k * k * Cin * Cout, device=dev).to(torch.bfloat16)
    dw = torch.empty(k * k * Cin * Cout, device=dev)
    gamma, beta = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    y, saved = ops.bn_fwd(xbn, gamma, beta, torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev), 0.1, 1e-5, True, None)
    req = dict(x=xbn, saved=saved, relu=True, relu_bits=None)

    def plain():
      ops.conv_bwd(d, x, dy, w, dw, need_dx=True)

    def fused():
      ops.conv_bwd(d, x, dy, w, dw, need_dx=True, bn_fuse=req)
    dx = ops.conv_bwd(d, x, dy, w, dw, need_dx=True)
    dg, db = torch.empty(Cin, device=dev), torch.empty(Cin, device=dev)

    def bn_plain():
      ops.bn_bwd(xbn, None, dx, gamma, saved, True, dg, db)
    fused()
    part = req['partials']

    def bn_fused():
      ops.bn_bwd(xbn, None, dx, gamma, saved, True, dg, db, partials=part)
    print('%-5s conv_bwd %6.1f us | with reductions %6.1f us | bn_bwd %6.1f us | bn_bwd given partials %6.1f us' %
          (name, timeit(plain), timeit(fused), timeit(bn_plain), timeit(bn_fused)), flush=True)


if __name__ == '__main__':
  main()
