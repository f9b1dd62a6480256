#!/usr/bin/env python3
"""Phase timeline of the row-streaming body (rowstream.hpp) on one layer (development; library built with -DRIGL_RS_TRACE:
tools/build_alt.sh trace conv -DRIGL_RS_TRACE, run with RIGL_HIP_LIB=build/alt/librigl_trace.so).  Stamps of wave 0 of every
workgroup: 0 entry; 1 filter landed + barrier; per fragment i: 2+3i top, 3+3i MFMAs issued, 4+3i epilogue issued; 62 loop done;
63 exit.  Usage: rs_trace.py H CIN COUT fwd|dgrad|dgrad_add"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import ops  # noqa: E402

dev = 'cuda:0'
H, Ci, Co, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
N = 128
ops.tune_set('rowstream', 2)
x = torch.randn(N, H, H, Ci, device=dev).to(torch.bfloat16)
dy = torch.randn(N, H, H, Co, device=dev).to(torch.bfloat16)
add = torch.randn(N, H, H, Ci, device=dev).to(torch.bfloat16)
w = (torch.randn(Ci * Co, device=dev) * 0.05).to(torch.bfloat16)
d = ops.conv_desc(N, H, H, Ci, Co, 1, 1, 1, 0, 0, H, H)
G = 2048
trace = torch.zeros(G * 64, dtype=torch.int64, device=dev)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def run():
  if mode == 'fwd':
    ops.conv_fwd(d, x, w, stats=True)
  elif mode == 'dgrad':
    ops.conv_dgrad(d, dy, w)
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
  os.environ['RIGL_RS_TRACE_PTR'] = str(trace.data_ptr())
  run()
  torch.cuda.synchronize()
  os.environ['RIGL_RS_TRACE_PTR'] = '0'
  t = trace.cpu().numpy().reshape(G, 64).astype(np.int64)
  ok = t[:, 63] > 0
  t = t[ok]
  t0 = t[:, 0].min()
  print('%s %dx%d %d->%d  %s: workgroups %d; clock ticks relative to the first workgroup entry (100 MHz s_memtime? see span)' % (
      mode, H, H, Ci, Co, 'cold' if cold else 'warm', len(t)))
  print('  entry      min %7d mean %7.0f max %7d' % (t[:, 0].min() - t0, (t[:, 0] - t0).mean(), t[:, 0].max() - t0))
  print('  filter+bar      mean %7.0f  (since own entry)' % (t[:, 1] - t[:, 0]).mean())
  for i in range(20):
    a, b, c = 2 + 3 * i, 3 + 3 * i, 4 + 3 * i
    m = t[:, c] > 0
    if not m.any():
      break
    print('  frag %2d (%4d wgs): top at %7.0f  mfma-issue %6.0f  epilogue-issue %6.0f' % (
        i, int(m.sum()), (t[m, a] - t[m, 0]).mean(), (t[m, b] - t[m, a]).mean(), (t[m, c] - t[m, b]).mean()))
  print('  loop done  mean %7.0f max %7d ; exit mean %7.0f max %7d (since first entry: last exit %d)' % (
      (t[:, 62] - t[:, 0]).mean(), (t[:, 62] - t[:, 0]).max(), (t[:, 63] - t[:, 0]).mean(), (t[:, 63] - t[:, 0]).max(),
      t[:, 63].max() - t0))
