#!/usr/bin/env python3
"""Experiment: capture one non-update RigL step in a HIP graph and replay it (how much of the
step is inter-kernel dispatch latency?).  Development tool, not used by bench.py."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import sparse_optimizers, sparse_utils, train, variables
from rigl_amd.workloads import resnet50

dev = torch.device('cuda', 0)
g = variables.reset_default_graph(dev)
model = resnet50.ResNet50(g, seed=0)
np.random.seed(0)
sparse_utils.get_mask_init_fn(g.get_masks(), 'erdos_renyi_kernel', 0.8, {})()
inner = train.MomentumOptimizer(0.05, 0.9, use_nesterov=True, graph=g)
opt = sparse_optimizers.SparseRigLOptimizer(inner, 0, 25000, 100, drop_fraction=0.3, drop_fraction_anneal='cosine')
gs = g.get_or_create_global_step()
images, labels = resnet50.synthetic_batch(128, dev)

def step():
  loss = model.loss(images, labels, label_smoothing=0.1)
  opt.minimize(loss, gs)
  return loss

for _ in range(5):
  step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
  step()
torch.cuda.synchronize()
print('eager  %.3f ms/step' % ((time.perf_counter() - t0) / 30 * 1e3))

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
  for _ in range(3):
    step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
  step()
torch.cuda.synchronize()
for _ in range(5):
  graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
  graph.replay()
torch.cuda.synchronize()
print('graph  %.3f ms/step' % ((time.perf_counter() - t0) / 30 * 1e3))
