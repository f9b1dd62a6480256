#!/usr/bin/env python3
"""What float32 arithmetic alone does to a whole-network gradient comparison -- no kernel of this repo involved.

oracle/resnet_cpu.py's ResNet-50 (batch 8, N(0,1) images) is trained for three steps twice with stock torch-CPU ops,
once in float32 and once in float64 from the same initial values, and the loss and every layer's dense dW are compared:

  as-is            every conv layer's dW is ~1 % (l2) off after ONE backward pass -- a handful of the 10^7 ReLU inputs
                   land on the other side of zero in float32 and flip an element of the gradient's mask (the fully
                   connected layer, above every ReLU, agrees to 3e-6; the loss to 4e-9);
  --same-piece     the float64 run takes the float32 run's ReLU signs and max-pool selections (both runs evaluate the
                   same linear piece of the network): worst layer 2.6e-5 .. 4.3e-5 of max |dW|.

These are the figures tests/test_k1_fp32_gpu.py's tolerances are set against.  CPU only, about a minute.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (repo root)
from oracle.resnet_cpu import ResNet50CPU  # noqa: E402


def build(dtype):
  torch.manual_seed(0)
  m = ResNet50CPU(seed=0, dtype=dtype)
  for blk in m.blocks:
    m.bn[blk['c3'][1]][0].data.fill_(0.5)        # the zero-initialised gamma would switch the residual branches off
  return m


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--same-piece', action='store_true')
  ap.add_argument('--lr', type=float, default=0.01)
  ap.add_argument('--threads', type=int, default=16)
  ap.add_argument('--batch', type=int, default=8)
  a = ap.parse_args()
  torch.set_num_threads(a.threads)
  m32, m64 = build(torch.float32), build(torch.float64)
  with torch.no_grad():
    for w32, w64 in zip(m32.w, m64.w):
      w64.copy_(w32.double())
  x = torch.randn(a.batch, 3, 224, 224)
  y = torch.randint(0, 1000, (a.batch,))
  print('batch %d, %s' % (a.batch, 'same linear piece' if a.same_piece else 'as is'), flush=True)
  relu0, pool0 = F.relu, F.max_pool2d
  tape = []

  def relu_rec(t):
    o = relu0(t)
    tape.append(o > 0)
    return o

  def pool_rec(t, *args, **kw):
    o, idx = pool0(t, *args, return_indices=True, **kw)
    tape.append(idx)
    return o

  def relu_play(t):
    return t * tape.pop(0).to(t.dtype)

  def pool_play(t, *args, **kw):
    idx = tape.pop(0)
    return t.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)

  for step in range(3):
    if a.same_piece:
      F.relu, F.max_pool2d = relu_rec, pool_rec
    l32 = m32.train_step(x, y, lr=a.lr, keep_dense=True)
    if a.same_piece:
      F.relu, F.max_pool2d = relu_play, pool_play
    l64 = m64.train_step(x.double(), y, lr=a.lr, keep_dense=True)
    F.relu, F.max_pool2d = relu0, pool0
    mx = [np.abs(g32 - g64).max() / np.abs(g64).max() for g32, g64 in zip(m32.dense_grads, m64.dense_grads)]
    l2 = [np.linalg.norm((g32 - g64).ravel()) / np.linalg.norm(g64.ravel())
          for g32, g64 in zip(m32.dense_grads, m64.dense_grads)]
    print('step %d  loss %.9f / %.9f (rel %.1e)   dW worst layer: max-rel %.3g, l2-rel %.3g;  fc layer max-rel %.3g'
          % (step, l32, l64, abs(l32 - l64) / abs(l64), max(mx), max(l2), mx[-1]), flush=True)


if __name__ == '__main__':
  main()
