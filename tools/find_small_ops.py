#!/usr/bin/env python3
"""Development probe: which Python lines of the ResNet-50 step launch the small framework kernels (fills, copies, casts)."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rigl_amd import sparse_optimizers, train, variables  # noqa: E402


def main():
  dev = torch.device('cuda', 0)
  g = variables.reset_default_graph(dev)
  wl = bench.build_workload('resnet50', g, dev, 128, 0, None)
  inner = train.MomentumOptimizer(wl['lr'](128), 0.9, use_nesterov=True, graph=g)
  opt = sparse_optimizers.SparseRigLOptimizer(inner, grow_init='zeros', initial_acc_scale=0.0, **wl['opt'])
  gs = g.get_or_create_global_step()

  def step():
    loss = wl['loss']()
    opt.minimize(loss, gs)

  for _ in range(3):
    step()
  gs.value = 10
  torch.cuda.synchronize()
  from torch.profiler import profile, ProfilerActivity
  with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    for _ in range(2):
      step()
  torch.cuda.synchronize()
  want = ('aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::_to_copy', 'aten::zeros', 'aten::mul', 'aten::add', 'aten::sum', 'aten::mean',
          'aten::contiguous', 'aten::clone')
  agg = collections.Counter()
  for ev in prof.events():
    if ev.name in want:
      frames = [f for f in (ev.stack or []) if '/root/repo' in f or 'rigl_amd' in f or 'bench.py' in f or 'workloads' in f]
      key = (ev.name, str(ev.input_shapes)[:60], frames[0][-90:] if frames else '(no repo frame: autograd engine)')
      agg[key] += 1
  for (name, shp, fr), n in agg.most_common(40):
    print('%3d x %-16s %-62s %s' % (n, name, shp, fr))


if __name__ == '__main__':
  main()
