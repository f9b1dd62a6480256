#!/usr/bin/env python3
"""Times single conv layers (fwd / dgrad / wgrad / one-call bwd) with HIP events; dev tool for A/B runs of alternative
library builds (RIGL_HIP_LIB) and kernel-selection knobs.   usage: layer_probe.py [--batch B] name=H,W,Cin,Cout,k,s ..."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import ops  # noqa: E402

DEFAULT = ['g3c1=14,14,1024,256,1,1', 'g3c3=14,14,256,1024,1,1', 'g3c2=14,14,256,256,3,1', 'g4c2=7,7,512,512,3,1',
           'g2c2=28,28,128,128,3,1', 'g4c1=7,7,2048,512,1,1']


def timeit(fn, iters=20, warmup=5):
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
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=128)
  ap.add_argument('--what', default='fwd,dgrad,wgrad')
  ap.add_argument('layers', nargs='*')
  a = ap.parse_args()
  dev = 'cuda:0'
  out = []
  for spec in (a.layers or DEFAULT):
    name, v = spec.split('=')
    H, W, Cin, Cout, k, s = [int(t) for t in v.split(',')]
    p = (k - 1) // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    N = a.batch
    d = ops.conv_desc(N, H, W, Cin, Cout, k, k, s, p, p, Ho, Wo)
    x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(N, Ho, Wo, Cout, device=dev).to(torch.bfloat16)
    w = torch.randn(k * k * Cin * Cout, device=dev).to(torch.bfloat16)
    y = torch.empty(N, Ho, Wo, Cout, device=dev, dtype=torch.bfloat16)
    dx = torch.empty(N, H, W, Cin, device=dev, dtype=torch.bfloat16)
    dw = torch.empty(k * k * Cin * Cout, device=dev, dtype=torch.float32)
    r = [name]
    if 'fwd' in a.what:
      r.append('fwd %6.1f' % timeit(lambda: ops.conv_fwd(d, x, w, y)))
    if 'dgrad' in a.what:
      r.append('dgrad %6.1f' % timeit(lambda: ops.conv_dgrad(d, dy, w, dx)))
    if 'wgrad' in a.what:
      r.append('wgrad %6.1f' % timeit(lambda: ops.conv_wgrad(d, x, dy, dw)))
    if 'bwd' in a.what:
      def f():
        ops.conv_bwd(d, x, dy, w, dw, need_dx=True)
      r.append('bwd %6.1f' % timeit(f))
    out.append(' '.join(r))
  print(' | '.join(out), flush=True)


if __name__ == '__main__':
  main()
