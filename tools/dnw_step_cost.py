#!/usr/bin/env python3
"""Per-step cost of the mask selection under DNW: Discovering Neural Wirings re-derives EVERY mask
EVERY step as the top-k of |W| (rigl/sparse_optimizers.py:341-480), so rigl_topk_mask_batched is not amortised over an
update period as RigL's prune/regrow is.  Times the ResNet-50 step (batch 128, ERK 0.8) under SparseDNWOptimizer and,
with events around it, the selection part of apply_gradients (|W| + the batched top-k + the shadow re-pack it forces)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--steps', type=int, default=30)
  ap.add_argument('--batch', type=int, default=128)
  a = ap.parse_args()
  from rigl_amd import ops, sparse_optimizers, sparse_utils, train, variables
  from rigl_amd.workloads import resnet50
  dev = torch.device('cuda', 0)
  g = variables.reset_default_graph(dev)
  model = resnet50.ResNet50(g, prune_first_layer=True, seed=0)
  np.random.seed(0)
  sparse_utils.get_mask_init_fn(g.get_masks(), 'erdos_renyi_kernel', 0.8, {})()
  images, labels = resnet50.synthetic_batch(a.batch, dev, seed=1234)
  inner = train.MomentumOptimizer(0.05, 0.9, use_nesterov=True, graph=g)
  opt = sparse_optimizers.SparseDNWOptimizer(inner, 0.8, 'erdos_renyi_kernel')
  gs = g.get_or_create_global_step()
  sel = []
  topk0 = ops.topk_mask_batched

  def timed_topk(items):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    topk0(items)
    e.record()
    sel.append((s, e))
  ops.topk_mask_batched = timed_topk

  def step():
    loss = model.loss(images, labels, label_smoothing=0.1)
    opt.minimize(loss, gs)

  for _ in range(8):
    step()
  torch.cuda.synchronize()
  sel.clear()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(a.steps):
    step()
  e.record()
  torch.cuda.synchronize()
  ms = s.elapsed_time(e) / a.steps
  topk_ms = sum(x.elapsed_time(y) for x, y in sel) / len(sel)
  n = sum(m.numel for m in g.get_masks())
  print('ResNet-50 DNW step, batch %d: %.3f ms per step (%.0f images/s); rigl_topk_mask_batched over %d masked weights '
        '(54 layers, one call): %.3f ms per step = %.1f %% of the step, %.0f GB/s of its 4.125 B/weight'
        % (a.batch, ms, a.batch / ms * 1e3, n, topk_ms, 100 * topk_ms / ms, n * 4.125 / topk_ms / 1e6))


if __name__ == '__main__':
  main()
