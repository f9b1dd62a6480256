#!/usr/bin/env python3
"""Why are the K1 kernels 20-40 % slower inside the training step than alone, even with cold operands?
One layer's one-call backward is timed by events around the call itself, 40 times per setting, with different kernels
enqueued right BEFORE it:
  alone-warm   the same operands every call          alone-cold   operands rotated through 768 MB
  after-stream a 400 MB element-wise copy before every call (what a batch-norm apply pass leaves behind: dirty lines in L2 /
               Infinity Cache, HBM write queues full)
  after-mfma   three MFMA-bound 3x3 forwards before every call (clock / power state)
  after-both   both
Usage: instep_probe.py [H CIN COUT]   (default 14 1024 256).  Development tool."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import ops  # noqa: E402

dev = 'cuda:0'
H, Ci, Co = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (14, 1024, 256)
N = 128
set_bytes = 2 * N * H * H * (2 * Ci + Co) * 2
copies = max(2, -(-768 * (1 << 20) // set_bytes))
xs = [torch.randn(N, H, H, Ci, device=dev).to(torch.bfloat16) for _ in range(copies)]
dys = [torch.randn(N, H, H, Co, device=dev).to(torch.bfloat16) for _ in range(copies)]
adds = [torch.randn(N, H, H, Ci, device=dev).to(torch.bfloat16) for _ in range(copies)]
w = (torch.randn(Ci * Co, device=dev) * 0.05).to(torch.bfloat16)
dw = torch.empty(Ci * Co, device=dev, dtype=torch.float32)
d = ops.conv_desc(N, H, H, Ci, Co, 1, 1, 1, 0, 0, H, H)
big_a = torch.randn(100 << 20, device=dev).to(torch.bfloat16)          # 200 MB
big_b = torch.empty_like(big_a)
d3 = ops.conv_desc(N, 14, 14, 256, 256, 3, 3, 1, 1, 1, 14, 14)
x3 = torch.randn(N, 14, 14, 256, device=dev).to(torch.bfloat16)
w3 = (torch.randn(9 * 256 * 256, device=dev) * 0.02).to(torch.bfloat16)
y3 = torch.empty(N, 14, 14, 256, device=dev, dtype=torch.bfloat16)


def pre_stream():
  big_b.copy_(big_a)


def pre_mfma():
  for _ in range(3):
    ops.conv_fwd(d3, x3, w3, y3)


def measure(pre, cold, iters=40):
  ts = []
  for i in range(iters + 5):
    j = i % copies if cold else 0
    if pre:
      for p in pre:
        p()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.conv_bwd(d, xs[j], dys[j], w, dw, need_dx=True, addend=adds[j])
    e.record()
    torch.cuda.synchronize()
    if i >= 5:
      ts.append(s.elapsed_time(e) * 1e3)
  ts.sort()
  return ts[len(ts) // 2], ts[0], ts[-1]


print('one-call backward of %dx%d %d->%d, batch %d: median / min / max us per call (events around the call)' % (H, H, Ci, Co, N))
for name, pre, cold in (('alone-warm', None, False), ('alone-cold', None, True), ('after-stream', [pre_stream], True),
                        ('after-mfma', [pre_mfma], True), ('after-both', [pre_mfma, pre_stream], True),
                        ('after-stream-warm', [pre_stream], False)):
  m, lo, hi = measure(pre, cold)
  print('  %-18s %7.1f  %7.1f  %7.1f' % (name, m, lo, hi), flush=True)
