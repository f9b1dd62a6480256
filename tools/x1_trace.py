#!/usr/bin/env python3
"""Phase timeline of the x1x1 forward on the 14x14 256 -> 1024 layer (development; library built with -DRIGL_X1_TRACE).
Stamps of wave 0: 0 entry; per chunk c: 1+6c top, 2+6c own loads/stores drained, 3+6c barrier passed, 4+6c MFMAs done,
5+6c outputs stored, 6+6c statistics done."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import ops  # noqa: E402

dev = 'cuda:0'
N, H, Ci, Co = 128, 14, 256, 1024
x = torch.randn(N, H, H, Ci, device=dev).to(torch.bfloat16)
y = torch.empty(N, H, H, Co, device=dev, dtype=torch.bfloat16)
w = (torch.randn(Ci * Co, device=dev) * 0.05).to(torch.bfloat16)
d = ops.conv_desc(N, H, H, Ci, Co, 1, 1, 1, 0, 0, H, H)
trace = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
for _ in range(3):
  ops.conv_fwd(d, x, w, y, stats=True)
torch.cuda.synchronize()
os.environ['RIGL_X1_TRACE_PTR'] = str(trace.data_ptr())
ops.conv_fwd(d, x, w, y, stats=True)
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(256, 64).astype(np.int64)
names = ['top', 'drained', 'barrier', 'mfma', 'stored', 'stats']
rel = t - t[:, :1]
ok = t[:, 1] > 0
print('workgroups with stamps', int(ok.sum()))
for c in range(10):
  row = []
  for k in range(6):
    i = 1 + c * 6 + k
    row.append('%s %6.0f' % (names[k], rel[ok, i].mean()))
  print('chunk %2d: ' % c + '  '.join(row))
