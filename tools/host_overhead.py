#!/usr/bin/env python3
"""Host-side (Python / ctypes / autograd) cost per layer call: tiny shapes so the kernels are
negligible, many iterations, one synchronize at the end.  Development tool."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import ops, pruning_layers as PL, variables as V  # noqa: E402
from rigl_amd.workloads import nn as gnn  # noqa: E402


def main():
  dev = 'cuda:0'
  g = V.reset_default_graph(dev)
  conv = PL.MaskedConv2d(g, 'c', 16, 16, (1, 1), sparsity_technique='threshold')
  bn = gnn.BatchNorm(g, 'bn', 16)
  g.finalize()
  x = torch.randn(2, 4, 4, 16, device=dev).to(torch.bfloat16).requires_grad_(True)
  n = 300

  def t(fn, label):
    for _ in range(20):
      fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
      fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    print('%-44s %7.1f us / call' % (label, dt / n * 1e6))

  d = conv.desc_for(2, 4, 4)
  y = torch.empty(2, 4, 4, 16, device=dev, dtype=torch.bfloat16)
  dw = torch.empty(16 * 16, device=dev)
  t(lambda: ops.conv_fwd(d, x.detach(), conv.vars.ohwi, y), 'ops.conv_fwd (preallocated y)')
  t(lambda: ops.conv_fwd(d, x.detach(), conv.vars.ohwi, stats=True), 'ops.conv_fwd (alloc y, stats)')
  t(lambda: ops.conv_dgrad(d, y, conv.vars.hwio), 'ops.conv_dgrad')
  t(lambda: ops.conv_wgrad(d, x.detach(), y, dw), 'ops.conv_wgrad')
  t(lambda: conv(x, True), 'MaskedConv2d.__call__ (fwd, autograd)')
  t(lambda: bn(conv(x, True), True, relu=True), 'conv + BatchNorm fwd')

  def fb():
    out = bn(conv(x, True), True, relu=True)
    out.float().sum().backward()
  t(fb, 'conv + BN fwd + bwd (incl. sum/cast)')
  t(lambda: ops._stream(), 'ops._stream()')
  t(lambda: torch.empty(2, 4, 4, 16, device=dev, dtype=torch.bfloat16), 'torch.empty')


if __name__ == '__main__':
  main()
