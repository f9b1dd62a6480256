#!/usr/bin/env python3
"""A/B of the channel-sliced single-pass 1x1 backward (bwdslice.hpp) against the shared-launch bodies it replaces, on the
1x1 / stride-1 "reduce" shapes of ResNet-50 at batch 128, in one process (rigl_tune_set("bwdslice", 0 | 1)), operands
rotated through 768 MB so that every call reads HBM.  Prints us per call.  Development tool."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import ops  # noqa: E402


def timeit(fn, iters, warmup=3):
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


dev = 'cuda:0'
N = int(os.environ.get('BS_BATCH', '128'))
SHAPES = ((14, 1024, 256), (28, 512, 128), (28, 512, 256), (56, 256, 128), (14, 1024, 512), (7, 2048, 512), (56, 256, 64))
for (H, Ci, Co) in SHAPES:
  set_bytes = 2 * N * H * H * (2 * Ci + Co) * 2
  copies = max(2, -(-768 * (1 << 20) // set_bytes))
  xs = [torch.randn(N, H, H, Ci, device=dev).to(torch.bfloat16) for _ in range(copies)]
  dys = [torch.randn(N, H, H, Co, device=dev).to(torch.bfloat16) for _ in range(copies)]
  adds = [torch.randn(N, H, H, Ci, device=dev).to(torch.bfloat16) for _ in range(copies)]
  w = (torch.randn(Ci * Co, device=dev) * 0.05).to(torch.bfloat16)
  dw = torch.empty(Ci * Co, device=dev, dtype=torch.float32)
  turn = [0]

  def nxt():
    turn[0] = (turn[0] + 1) % copies
    return turn[0]
  for v in (0, 1):
    ops.tune_set('bwdslice', v)
    d = ops.conv_desc(N, H, H, Ci, Co, 1, 1, 1, 0, 0, H, H)
    it = max(20, copies)
    t_d = timeit(lambda: (lambda i: ops.conv_dgrad(d, dys[i], w, addend=adds[i]))(nxt()), it)
    t_b = timeit(lambda: (lambda i: ops.conv_bwd(d, xs[i], dys[i], w, dw, need_dx=True, addend=adds[i]))(nxt()), it)
    t_b0 = timeit(lambda: (lambda i: ops.conv_bwd(d, xs[i], dys[i], w, dw, need_dx=True))(nxt()), it)
    gb = N * H * H * (3 * Ci + Co) * 2 / 1e3
    print('%2dx%2d %4d->%4d bwdslice=%d  dgrad+add %6.1f us  bwd+add %6.1f us (%4.0f GB/s algorithmic)  bwd %6.1f us' % (
        H, H, Ci, Co, v, t_d, t_b, gb / t_b, t_b0), flush=True)
  ops.tune_unset('bwdslice')
  del xs, dys, adds
