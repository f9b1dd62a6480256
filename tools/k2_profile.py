#!/usr/bin/env python3
"""Runs the whole-ResNet-50 prune/regrow (K2) a few times -- for a per-kernel rocprofv3 table:

  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o k2 -- python tools/k2_profile.py
  python tools/k2_profile.py --summarise OUT/k2_kernel_stats.csv
"""
import argparse
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--iters', type=int, default=10)
  ap.add_argument('--sparsity', type=float, default=0.8)
  ap.add_argument('--summarise', default=None)
  a = ap.parse_args()
  if a.summarise:
    rows = [r for r in csv.DictReader(open(a.summarise)) if 'rigl::k2::' in r['Name']]
    tot = 0.0
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
      per = float(r['TotalDurationNs']) / a.iters / 1e3
      tot += per
      print('%-60s calls/update %4.1f  avg %7.1f us  per update %7.1f us' % (
          r['Name'].replace('rigl::k2::', '').replace('void ', '')[:60], int(r['Calls']) / a.iters,
          float(r['AverageNs']) / 1e3, per))
    print('K2 kernels per update: %.1f us' % tot)
    return
  import numpy as np
  import torch
  from rigl_amd import ops
  from rigl_amd.workloads import shapes as layer_shapes
  dev = torch.device('cuda', 0)
  torch.manual_seed(0)
  layers = []
  for sh in layer_shapes.resnet50_masks().values():
    n = int(np.prod(sh))
    layers.append(dict(w=torch.randn(n, device=dev) * 0.05, momentum=torch.zeros(n, device=dev),
                       dense_grad=torch.randn(n, device=dev) * 1e-3,
                       mask_bits=ops.mask_pack((torch.rand(n, device=dev) < 1.0 - a.sparsity).float())))
  for _ in range(a.iters):
    ops.prune_regrow(layers, 0.3)
  torch.cuda.synchronize()


if __name__ == '__main__':
  main()
