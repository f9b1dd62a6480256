#!/usr/bin/env python3
"""Phase timeline of the persistent c3x3 forward (development): needs a library built with -DRIGL_C3_TRACE
(tools/build_alt.sh trace conv -DRIGL_C3_TRACE; RIGL_HIP_LIB=build/alt/librigl_trace.so).  Wave 0 of every workgroup stamps
s_memtime at: 0 entry, 1 filter loaded, then per tile t: 2+5t loop top, 3+5t own loads/stores drained, 4+5t barrier passed,
5+5t next patch issued, 6+5t M-tiles done; 63 exit."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import ops  # noqa: E402

dev = 'cuda:0'
N, H, C = 128, 56, 64
x = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16)
y = torch.empty_like(x)
w = (torch.randn(9 * C * C, device=dev) * 0.05).to(torch.bfloat16)
d = ops.conv_desc(N, H, H, C, C, 3, 3, 1, 1, 1, H, H)
trace = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
for _ in range(3):
  ops.conv_fwd(d, x, w, y, stats=True)
torch.cuda.synchronize()
os.environ['RIGL_C3_TRACE_PTR'] = str(trace.data_ptr())
ops.conv_fwd(d, x, w, y, stats=True)
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(256, 64).astype(np.int64)
t0 = t[:, 0].min()
names = {0: 'entry', 1: 'filter in regs', 63: 'exit'}
for it in range(4):
  names.update({2 + 5 * it: 'tile %d top' % it, 3 + 5 * it: 'tile %d vmcnt(0)' % it, 4 + 5 * it: 'tile %d barrier' % it,
                5 + 5 * it: 'tile %d next patch issued' % it, 6 + 5 * it: 'tile %d M-tiles done' % it})
prev = None
for i in sorted(names):
  col = t[:, i]
  ok = col > 0
  if not ok.any():
    continue
  rel = (col[ok] - t0)
  line = '%-26s mean %8.0f  min %8.0f  max %8.0f ticks' % (names[i], rel.mean(), rel.min(), rel.max())
  if prev is not None:
    line += '   (+%.0f)' % (rel.mean() - prev)
  prev = rel.mean()
  print(line)
