#!/usr/bin/env python3
"""Phase breakdown of the channel-sliced single-pass backward (bwdslice.hpp) on one layer (development; library built with
-DRIGL_BS_TRACE: tools/build_alt.sh bstrace conv -DRIGL_BS_TRACE, run with RIGL_HIP_LIB=build/alt/librigl_bstrace.so).
s_memtime ticks summed over the K-tiles of a workgroup, waves 0 (dgrad on even tiles) and 4 (odd tiles):
0 wait + barrier, 1 DMA issue, 2 flush of the previous dX tile, 3 addend load, 4 dgrad, 5 weight gradient, 6 staging write,
7 prologue + epilogue.  Usage: bs_trace.py H CIN COUT"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import ops  # noqa: E402

dev = 'cuda:0'
H, Ci, Co = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = 128
x = torch.randn(N, H, H, Ci, device=dev).to(torch.bfloat16)
dy = torch.randn(N, H, H, Co, device=dev).to(torch.bfloat16)
add = torch.randn(N, H, H, Ci, device=dev).to(torch.bfloat16)
w = (torch.randn(Ci * Co, device=dev) * 0.05).to(torch.bfloat16)
dw = torch.empty(Ci * Co, device=dev, dtype=torch.float32)
d = ops.conv_desc(N, H, H, Ci, Co, 1, 1, 1, 0, 0, H, H)
G = 256
trace = torch.zeros(G * 16, dtype=torch.int64, device=dev)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
names = ['wait+barrier', 'dma issue', 'flush', 'addend load', 'dgrad', 'wgrad', 'staging', 'pro+epilogue']
for mode in ('bwd+add', 'bwd', 'dgrad+add'):
  def run():
    if mode == 'bwd+add':
      ops.conv_bwd(d, x, dy, w, dw, need_dx=True, addend=add)
    elif mode == 'bwd':
      ops.conv_bwd(d, x, dy, w, dw, need_dx=True)
    else:
      ops.conv_dgrad(d, dy, w, addend=add)
  for _ in range(3):
    run()
  torch.cuda.synchronize()
  for cold in (0, 1):
    trace.zero_()
    if cold:
      flush.fill_(1)
    torch.cuda.synchronize()
    os.environ['RIGL_BS_TRACE_PTR'] = str(trace.data_ptr())
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    run()
    e.record()
    torch.cuda.synchronize()
    os.environ['RIGL_BS_TRACE_PTR'] = '0'
    t = trace.cpu().numpy().reshape(G, 2, 8).astype(np.float64)
    tot = t.sum(2)
    print('%s %dx%d %d->%d %s: call %.1f us (with stamps); ticks per workgroup wave0 %.0f wave4 %.0f' % (
        mode, H, H, Ci, Co, 'cold' if cold else 'warm', s.elapsed_time(e) * 1e3, tot[:, 0].mean(), tot[:, 1].mean()))
    for i, n in enumerate(names):
      print('   %-14s wave0 %8.0f (%4.1f%%)   wave4 %8.0f (%4.1f%%)' % (
          n, t[:, 0, i].mean(), 100 * t[:, 0, i].mean() / tot[:, 0].mean(), t[:, 1, i].mean(), 100 * t[:, 1, i].mean() / tot[:, 1].mean()))
