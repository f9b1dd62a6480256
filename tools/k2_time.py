#!/usr/bin/env python3
"""Times one whole-ResNet-50 mask update (K2) with HIP events: sparsity 0.8 / 0.99, with and without a noise tensor.
Development tool; RIGL_K2_V1=1 selects the 23-launch pipeline for comparison."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import ops  # noqa: E402
from rigl_amd.workloads import shapes as layer_shapes  # noqa: E402


def main():
  dev = torch.device('cuda', 0)
  out = []
  for sparsity in (0.8, 0.99):
    for noise in (False, True):
      torch.manual_seed(0)
      layers = []
      for sh in layer_shapes.resnet50_masks().values():
        n = int(np.prod(sh))
        d = dict(w=torch.randn(n, device=dev) * 0.05, momentum=torch.zeros(n, device=dev),
                 dense_grad=torch.randn(n, device=dev) * 1e-3,
                 mask_bits=ops.mask_pack((torch.rand(n, device=dev) < 1.0 - sparsity).float()))
        if noise:
          d['drop_noise'] = torch.randn(n, device=dev) * 1e-5
        layers.append(d)
      n_tot = sum(l['w'].numel() for l in layers)
      for _ in range(3):
        ops.prune_regrow(layers, 0.3)
      torch.cuda.synchronize()
      s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      s.record()
      for _ in range(20):
        ops.prune_regrow(layers, 0.3)
      e.record()
      torch.cuda.synchronize()
      us = s.elapsed_time(e) / 20 * 1e3
      bpw = 12.25 if noise else 8.25
      out.append('sparsity %.2f noise %d: %7.1f us  %6.0f GB/s algorithmic (%.2f B/weight)' % (sparsity, noise, us, bpw * n_tot / us / 1e3, bpw))
  print('\n'.join(out))


if __name__ == '__main__':
  main()
