#!/usr/bin/env python3
"""The batch-norm apply + ReLU on the operand load of the row-streaming forward (rigl_masked_conv2d_fwd_bnrelu) against the
two calls it replaces (rigl_bn_fwd_stats: finalize + apply, then rigl_masked_conv2d_fwd_stats), per conv3 shape of ResNet-50
at batch 128, operands rotated through 768 MB; and bit-identity of y, the activated tensor and the statistics parts.
Development tool (the step-level A/B is tools/ab.sh with RIGL_BN_ON_LOAD=0 / 1)."""
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
for (H, Ci, Co) in ((56, 64, 256), (28, 128, 512), (14, 256, 1024), (56, 256, 64), (28, 512, 128)):
  d = ops.conv_desc(N, H, H, Ci, Co, 1, 1, 1, 0, 0, H, H)
  if not ops.conv_fwd_takes_bn_input(d):
    print('%2dx%2d %4d->%4d  not taken' % (H, H, Ci, Co))
    continue
  M = N * H * H
  set_bytes = M * (2 * Ci + Co) * 2
  copies = max(2, -(-768 * (1 << 20) // set_bytes))
  xs = [torch.randn(N, H, H, Ci, device=dev).to(torch.bfloat16) for _ in range(copies)]
  a2s = [torch.empty(N, H, H, Ci, device=dev, dtype=torch.bfloat16) for _ in range(copies)]
  ys = [torch.empty(N, H, H, Co, device=dev, dtype=torch.bfloat16) for _ in range(copies)]
  w = (torch.randn(Ci * Co, device=dev) * 0.05).to(torch.bfloat16)
  gamma, beta = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.1
  rm, rv = torch.zeros(Ci, device=dev), torch.ones(Ci, device=dev)
  part = torch.stack([xs[0].float().view(-1, Ci).sum(0), (xs[0].float() ** 2).view(-1, Ci).sum(0)])[None].contiguous()
  turn = [0]

  def nxt():
    turn[0] = (turn[0] + 1) % copies
    return turn[0]

  def two_calls(i):
    a2, saved = ops.bn_fwd(xs[i], gamma, beta, rm, rv, 0.1, 1e-5, True, None, partials=part)
    return ops.conv_fwd(d, a2, w, ys[i], stats=True), a2

  def one_call(i):
    saved = ops.bn_statistics(xs[i], gamma, beta, rm, rv, 0.1, 1e-5, partials=part)
    return ops.conv_fwd_bnrelu(d, xs[i], saved, w, a2s[i], ys[i], stats=True), a2s[i]
  it = max(10, copies)
  t2 = timeit(lambda: two_calls(nxt()), it)
  t1 = timeit(lambda: one_call(nxt()), it)
  t_conv = timeit(lambda: (lambda i: ops.conv_fwd(d, xs[i], w, ys[i], stats=True))(nxt()), it)
  (y2, p2), a2 = two_calls(0)
  y2, p2, a2 = y2.clone(), p2.clone(), a2.clone()
  (y1, p1), a1 = one_call(0)
  torch.cuda.synchronize()
  same = (torch.equal(y1.view(torch.int16), y2.view(torch.int16)), torch.equal(a1.view(torch.int16), a2.view(torch.int16)),
          torch.equal(p1, p2))
  print('%2dx%2d %4d->%4d  finalize + apply + conv %6.1f us   finalize + conv with the apply on its load %6.1f us   (conv alone %5.1f)   '
        'bit-identical y / activated / statistics: %s' % (H, H, Ci, Co, t2, t1, t_conv, same), flush=True)
