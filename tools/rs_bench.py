#!/usr/bin/env python3
"""A/B of the row-streaming 1x1 body (rowstream.hpp) against the bodies it replaces on every 1x1 / stride-1 layer shape of
ResNet-50 at batch 128, in one process (rigl_tune_set("rowstream", 0 | 2)), operands rotated through 768 MB so that every
call reads HBM.  Prints us per call and the algorithmic GB/s of the forward.  Development tool."""
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
N = int(os.environ.get('RS_BATCH', '128'))
SHAPES = ((56, 64, 256), (56, 64, 64), (56, 256, 64), (56, 256, 128), (28, 128, 512), (28, 512, 128), (28, 512, 256),
          (14, 256, 1024), (14, 1024, 256), (14, 1024, 512), (7, 512, 2048), (7, 2048, 512))
only = os.environ.get('RS_ONLY')
for (H, Ci, Co) in SHAPES:
  if only and only != '%d_%d_%d' % (H, Ci, Co):
    continue
  set_bytes = 2 * N * H * H * (Ci + Co) * 2
  copies = max(2, -(-768 * (1 << 20) // set_bytes))
  xs = [torch.randn(N, H, H, Ci, device=dev).to(torch.bfloat16) for _ in range(copies)]
  dys = [torch.randn(N, H, H, Co, device=dev).to(torch.bfloat16) for _ in range(copies)]
  ys = [torch.empty(N, H, H, Co, device=dev, dtype=torch.bfloat16) for _ in range(copies)]
  adds = [torch.randn(N, H, H, Ci, device=dev).to(torch.bfloat16) for _ in range(copies)]
  w = (torch.randn(Ci * Co, device=dev) * 0.05).to(torch.bfloat16)
  dw = torch.empty(Ci * Co, device=dev, dtype=torch.float32)
  turn = [0]

  def nxt():
    turn[0] = (turn[0] + 1) % copies
    return turn[0]
  for v in (0, 2):
    ops.tune_set('rowstream', v)
    d = ops.conv_desc(N, H, H, Ci, Co, 1, 1, 1, 0, 0, H, H)
    it = max(10, copies)
    t_f = timeit(lambda: (lambda i: ops.conv_fwd(d, xs[i], w, ys[i], stats=True))(nxt()), it)
    t_d = timeit(lambda: (lambda i: ops.conv_dgrad(d, dys[i], w, addend=adds[i]))(nxt()), it)
    t_d0 = timeit(lambda: (lambda i: ops.conv_dgrad(d, dys[i], w))(nxt()), it)
    t_b = timeit(lambda: (lambda i: ops.conv_bwd(d, xs[i], dys[i], w, dw, need_dx=True, addend=adds[i]))(nxt()), it)
    gb_f = N * H * H * (Ci + Co) * 2 / 1e3
    gb_d = N * H * H * (2 * Ci + Co) * 2 / 1e3
    print('%2dx%2d %4d->%4d rowstream=%d  fwd+stats %6.1f us (%4.0f GB/s)  dgrad+add %6.1f (%4.0f GB/s)  dgrad %6.1f  bwd+add %6.1f us' % (
        H, H, Ci, Co, v, t_f, gb_f / t_f, t_d, gb_d / t_d, t_d0, t_b), flush=True)
  ops.tune_unset('rowstream')
  del xs, dys, ys, adds
