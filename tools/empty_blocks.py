#!/usr/bin/env python3
"""could the mask pay for something on the non-update steps of ResNet-50 at ERK 0.99?  Counts, per masked
layer, the blocks of the [kh * kw * cin] x [cout] weight matrix (the GEMM's K x N operand; flat HWIO index space) whose mask
is ENTIRELY zero -- the only blocks a dense-storage MFMA kernel could skip (a K-tile of the forward / dgrad reduction, an
output block of the weight gradient) -- for 32 x 32 and 64 x 64 blocks, masks drawn as the benchmark draws them
(sparse_utils.get_mask_init_fn 'erdos_renyi_kernel', np.random.seed(0), stem dense).  CPU only.
  python tools/empty_blocks.py [--sparsity 0.99]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import sparse_utils  # noqa: E402
from rigl_amd.workloads import shapes as layer_shapes  # noqa: E402


class _Mask:
  def __init__(self, name, shape):
    self.name, self.shape = name, tuple(shape)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--sparsity', type=float, default=0.99)
  a = ap.parse_args()
  shapes = layer_shapes.resnet50_masks()
  names = list(shapes)
  masks = [_Mask(n, shapes[n]) for n in names]
  custom = {names[0]: 0.0} if a.sparsity >= 0.95 else {}      # (the 99 % configuration keeps the stem dense: README.md:17-20)
  sp = sparse_utils.get_sparsities(masks, 'erdos_renyi_kernel', a.sparsity, custom, extract_name_fn=lambda n: n)
  rng = np.random.RandomState(0)
  tot = {32: [0, 0.0], 64: [0, 0.0]}
  wtot = {32: 0.0, 64: 0.0}
  macs_all = 0.0
  print('%-58s %14s %8s | empty 32x32 | empty 64x64' % ('layer', 'shape', 'sparsity'))
  for n in names:
    sh = shapes[n]
    s = sp[n]
    m = sparse_utils.get_mask_random_numpy(sh, s, rng).reshape(-1, sh[-1])        # [kh*kw*cin][cout]
    row = '%-58s %14s %8.4f |' % (n[-58:], 'x'.join(map(str, sh)), s)
    for b in (32, 64):
      K, N = m.shape
      kb, nb = -(-K // b), -(-N // b)
      pad = np.zeros((kb * b, nb * b), m.dtype)
      pad[:K, :N] = m
      blk = pad.reshape(kb, b, nb, b).sum((1, 3))
      frac = float((blk == 0).mean())
      tot[b][0] += blk.size
      tot[b][1] += float((blk == 0).sum())
      wtot[b] += frac * m.size
      row += ' %10.2f%% |' % (100 * frac)
    macs_all += m.size
    print(row)
  for b in (32, 64):
    print('all layers, %dx%d blocks: %.2f%% of the blocks empty (%.2f%% weighted by layer size)' % (
        b, b, 100 * tot[b][1] / tot[b][0], 100 * wtot[b] / macs_all))


if __name__ == '__main__':
  main()
