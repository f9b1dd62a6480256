#!/usr/bin/env python3
"""A/B of the slab-resident 3x3 kernels (c3x3.hpp) against the generic bodies on one layer shape, in one process
(rigl_tune_set("c3x3", 0 | 1)): forward (+ statistics), dgrad, weight gradient, the one-call backward; operands either
re-used (cache-warm) or rotated through 768 MB (--cold, as inside the training step).  Development tool."""
import argparse
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


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=128)
  ap.add_argument('--hw', type=int, default=56)
  ap.add_argument('--iters', type=int, default=20)
  ap.add_argument('--knob', default='c3x3')
  a = ap.parse_args()
  dev = 'cuda:0'
  N, H, C = a.batch, a.hw, 64
  set_bytes = 4 * N * H * H * C * 2
  for cold in (False, True):
    copies = max(2, -(-768 * (1 << 20) // set_bytes)) if cold else 1
    xs = [torch.randn(N, H, H, C, device=dev).to(torch.bfloat16) for _ in range(copies)]
    dys = [torch.randn(N, H, H, C, device=dev).to(torch.bfloat16) for _ in range(copies)]
    ys = [torch.empty(N, H, H, C, device=dev, dtype=torch.bfloat16) for _ in range(copies)]
    w = (torch.randn(9 * C * C, device=dev) * 0.05).to(torch.bfloat16)
    dw = torch.empty(9 * C * C, device=dev, dtype=torch.float32)
    turn = [0]

    def nxt():
      turn[0] = (turn[0] + 1) % copies
      return turn[0]
    for v in (0, 1):
      ops.tune_set(a.knob, v)
      d = ops.conv_desc(N, H, H, C, C, 3, 3, 1, 1, 1, H, H)
      iters = max(a.iters, copies)
      t_f = timeit(lambda: (lambda i: ops.conv_fwd(d, xs[i], w, ys[i]))(nxt()), iters)
      t_s = timeit(lambda: (lambda i: ops.conv_fwd(d, xs[i], w, ys[i], stats=True))(nxt()), iters)
      t_d = timeit(lambda: (lambda i: ops.conv_dgrad(d, dys[i], w, ys[i]))(nxt()), iters)
      t_w = timeit(lambda: (lambda i: ops.conv_wgrad(d, xs[i], dys[i], dw))(nxt()), iters)
      t_b = timeit(lambda: (lambda i: ops.conv_bwd(d, xs[i], dys[i], w, dw, need_dx=True))(nxt()), iters)
      flops = 2.0 * N * H * H * C * C * 9
      print('%s %s=%d  fwd %6.1f us (%5.0f TF)  fwd+stats %6.1f  dgrad %6.1f  wgrad %6.1f  bwd %6.1f us' % (
          'cold' if cold else 'warm', a.knob, v, t_f, flops / t_f / 1e6, t_s, t_d, t_w, t_b), flush=True)
    ops.tune_unset(a.knob)
    del xs, dys, ys


if __name__ == '__main__':
  main()
