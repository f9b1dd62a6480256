#!/usr/bin/env python3
"""Development probe: what the batch-norm statistics epilogue costs the K1 forward, per ResNet-50 layer shape
(conv_fwd(stats=True) vs conv_fwd(), each kernel alone, batch 128)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import ops  # noqa: E402
from tools.bench_kernels import resnet50_convs, timeit  # noqa: E402


def main():
  dev = 'cuda:0'
  seen, tot0, tot1 = {}, 0.0, 0.0
  for (name, N, H, W, Cin, Cout, k, s, p, Ho, Wo) in resnet50_convs(128):
    if name in ('fc',):
      continue
    key = (H, W, Cin, Cout, k, s)
    if key not in seen:
      d = ops.conv_desc(N, H, W, Cin, Cout, k, k, s, p, p, Ho, Wo)
      x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
      w = torch.randn(k * k * Cin * Cout, device=dev).to(torch.bfloat16)
      t0 = timeit(lambda: ops.conv_fwd(d, x, w), 10)
      t1 = timeit(lambda: ops.conv_fwd(d, x, w, stats=True), 10)
      seen[key] = (t0, t1)
      print('%-10s %-28s plain %7.1f us  with statistics %7.1f us  (+%.1f)' % (name, key, t0 * 1e3, t1 * 1e3, (t1 - t0) * 1e3), flush=True)
    t0, t1 = seen[key]
    tot0 += t0
    tot1 += t1
  print('TOTAL forward over the 53 convs: plain %.3f ms, with statistics %.3f ms' % (tot0, tot1))


if __name__ == '__main__':
  main()
