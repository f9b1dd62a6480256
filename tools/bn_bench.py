#!/usr/bin/env python3
"""The batch-norm passes of the ResNet-50 step alone, per activation shape (batch 128), operands rotated through 768 MB:
forward (finalize from producer-style partials + apply [+ residual + ReLU + bits]) and backward (reduce + finalize + apply
[+ dres]), us per call and GB/s of the tensors each call touches.  Development tool (the per-kernel split is in the
rocprofv3 table of the step)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import ops  # noqa: E402

dev = 'cuda:0'
N = 128


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


for (H, C, res) in ((56, 64, False), (56, 256, True), (28, 128, False), (28, 512, True), (14, 256, False), (14, 1024, True),
                    (7, 512, False), (7, 2048, True)):
  M = N * H * H
  tb = M * C * 2
  copies = max(2, -(-768 * (1 << 20) // (tb * (4 if res else 3))))
  xs = [torch.randn(N, H, H, C, device=dev).to(torch.bfloat16) for _ in range(copies)]
  dys = [torch.randn(N, H, H, C, device=dev).to(torch.bfloat16) for _ in range(copies)]
  rs = [torch.randn(N, H, H, C, device=dev).to(torch.bfloat16) for _ in range(copies)] if res else None
  gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
  rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
  part = torch.randn((M + 127) // 128, 2, C, device=dev).abs()
  dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
  turn = [0]

  def nxt():
    turn[0] = (turn[0] + 1) % copies
    return turn[0]
  out = ops.bn_fwd(xs[0], gamma, beta, rm, rv, 0.1, 1e-5, True, rs[0] if res else None, partials=part, want_relu_bits=res)
  saved, bits = out[1], (out[2] if res else None)
  it = max(20, copies)
  t_f = timeit(lambda: (lambda i: ops.bn_fwd(xs[i], gamma, beta, rm, rv, 0.1, 1e-5, True, rs[i] if res else None, partials=part,
                                              want_relu_bits=res))(nxt()), it)
  t_b = timeit(lambda: (lambda i: ops.bn_bwd(xs[i], None, dys[i], gamma, saved, True, dg, db, want_dres=res, relu_bits=bits))(nxt()), it)
  nf, nb = (3 if res else 2), (6 if res else 5)
  print('%2dx%2d x %4d %s  forward %6.1f us (%4.0f GB/s over %d tensor passes)   backward %6.1f us (%4.0f GB/s over %d)' % (
      H, H, C, 'bn3+res' if res else 'bn1/2  ', t_f, nf * tb / t_f / 1e3, nf, t_b, nb * tb / t_b / 1e3, nb), flush=True)
  del xs, dys, rs
