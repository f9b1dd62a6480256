#!/usr/bin/env python3
"""Calibration: K1 against the vendor libraries on the ResNet-50 layer shapes (batch B per GPU).

For every distinct conv shape: this library's fwd / dgrad / wgrad kernels, MIOpen through
torch (bf16, channels_last, aten.convolution / convolution_backward with an output mask so
dgrad and wgrad are timed apart), and -- for the 1x1 stride-1 layers, which are plain GEMMs --
hipBLASLt through torch.matmul on the same [pixels, channels] matrices.  Development tool:
nothing in the product or in bench.py calls the vendor paths.  Writes gpurun_out/vendor_compare.json.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import ops  # noqa: E402
from tools.bench_kernels import resnet50_convs, timeit  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=128)
  ap.add_argument('--iters', type=int, default=10)
  ap.add_argument('--find', type=int, default=1, help='1: MIOpen find mode (cudnn.benchmark), 0: immediate mode')
  ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'vendor_compare.json'))
  a = ap.parse_args()
  torch.backends.cudnn.benchmark = bool(a.find)
  dev = 'cuda:0'
  rows, seen = [], {}
  tot = {}
  for (name, N, H, W, Cin, Cout, k, s, p, Ho, Wo) in resnet50_convs(a.batch):
    if name in ('stem', 'fc'):
      continue
    key = (H, W, Cin, Cout, k, s)
    macs = N * Ho * Wo * Cout * k * k * Cin
    if key not in seen:
      d = ops.conv_desc(N, H, W, Cin, Cout, k, k, s, p, p, Ho, Wo)
      x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
      dy = torch.randn(N, Ho, Wo, Cout, device=dev).to(torch.bfloat16)
      w = torch.randn(k * k * Cin * Cout, device=dev).to(torch.bfloat16)
      y = torch.empty(N, Ho, Wo, Cout, device=dev, dtype=torch.bfloat16)
      dx = torch.empty(N, H, W, Cin, device=dev, dtype=torch.bfloat16)
      dw = torch.empty(k * k * Cin * Cout, device=dev, dtype=torch.float32)
      r = dict(ours_fwd=timeit(lambda: ops.conv_fwd(d, x, w, y), a.iters),
               ours_dgrad=timeit(lambda: ops.conv_dgrad(d, dy, w, dx), a.iters),
               ours_wgrad=timeit(lambda: ops.conv_wgrad(d, x, dy, dw), a.iters))
      # MIOpen: NCHW-shaped channels_last views of the same NHWC memory, OIHW-shaped channels_last weights
      xv, dyv = x.permute(0, 3, 1, 2), dy.permute(0, 3, 1, 2)
      wv = torch.randn(Cout, Cin, k, k, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
      conv = lambda: torch.ops.aten.convolution(xv, wv, None, [s, s], [p, p], [1, 1], False, [0, 0], 1)
      bwd = lambda mask: torch.ops.aten.convolution_backward(dyv, xv, wv, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, mask)
      try:
        r['miopen_fwd'] = timeit(conv, a.iters)
        r['miopen_dgrad'] = timeit(lambda: bwd([True, False, False]), a.iters)
        r['miopen_wgrad'] = timeit(lambda: bwd([False, True, False]), a.iters)
      except Exception as e:  # pylint: disable=broad-except
        print('MIOpen failed', name, key, repr(e)[:200], flush=True)
      if k == 1 and s == 1:
        xm, dym = x.view(-1, Cin), dy.view(-1, Cout)
        wm = torch.randn(Cin, Cout, device=dev).to(torch.bfloat16)
        wt = wm.t().contiguous()
        r['blaslt_fwd'] = timeit(lambda: torch.matmul(xm, wm), a.iters)
        r['blaslt_dgrad'] = timeit(lambda: torch.matmul(dym, wt), a.iters)
        r['blaslt_wgrad'] = timeit(lambda: torch.matmul(xm.t(), dym), a.iters)
      seen[key] = r
      del x, dy, w, y, dx, dw
    r = seen[key]
    rows.append(dict(name=name, shape=key, macs=macs, **r))
    for kk, v in r.items():
      tot[kk] = tot.get(kk, 0.0) + v
    for lib in ('blaslt',):       # totals over the 1x1 layers only, for the like-for-like line
      if k == 1 and s == 1:
        for dirn in ('fwd', 'dgrad', 'wgrad'):
          tot['ours1x1_' + dirn] = tot.get('ours1x1_' + dirn, 0.0) + r['ours_' + dirn]
    tf = lambda t: 2 * macs / (t * 1e-3) / 1e12 if t else 0.0
    print('%-9s %-26s ' % (name, key) + ' | '.join(
        '%s %s' % (dirn, ' '.join('%s %.3f (%4.0f TF)' % (lib[:2], r[lib + '_' + dirn], tf(r[lib + '_' + dirn]))
                                  for lib in ('ours', 'miopen', 'blaslt') if lib + '_' + dirn in r))
        for dirn in ('fwd', 'dgrad', 'wgrad')), flush=True)
  print('TOTALS ms over the 52 convs (stem excluded):', {k: round(v, 3) for k, v in sorted(tot.items())}, flush=True)
  os.makedirs(os.path.dirname(a.out), exist_ok=True)
  with open(a.out, 'w') as f:
    json.dump(dict(batch=a.batch, find=a.find, rows=rows, totals=tot), f, indent=1)


if __name__ == '__main__':
  main()
