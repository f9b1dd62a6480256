#!/usr/bin/env python3
"""Per-layer IN-STEP K1 timings of the ResNet-50 training step: every K1 dispatch of N profiled
steps is stamped by its own start / stop events (rigl_prof_collect_launches) and tagged with its conv descriptor, so
the durations can be summed per layer shape and pass -- the same rows as tools/bench_kernels.py prints for the kernels
run alone, and the two are diffed here:

  python tools/bench_kernels.py --out gpurun_out/bk_warm.json            # each kernel alone, operands cache-warm
  python tools/bench_kernels.py --cold --out gpurun_out/bk_cold.json     # ... rotating through > 600 MB of operand copies
  python tools/instep_table.py --alone gpurun_out/bk_warm.json --cold gpurun_out/bk_cold.json

Columns (us per step, summed over the layers of a shape): forward / backward in the step, alone (warm), alone (cold)."""
import argparse
import collections
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--warmup', type=int, default=8)
  ap.add_argument('--batch', type=int, default=128)
  ap.add_argument('--alone', default=None, help='bench_kernels.py JSON (kernels alone, cache-warm)')
  ap.add_argument('--cold', default=None, help='bench_kernels.py --cold JSON')
  ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'instep_table.json'))
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
  opt = sparse_optimizers.SparseRigLOptimizer(inner, begin_step=0, end_step=25000, frequency=100000, drop_fraction=0.3,
                                              drop_fraction_anneal='cosine', grow_init='zeros', initial_acc_scale=0.0)
  gs = g.get_or_create_global_step()

  def step():
    loss = model.loss(images, labels, label_smoothing=0.1)
    opt.minimize(loss, gs)

  for _ in range(a.warmup):
    step()
  torch.cuda.synchronize()
  ops.prof_collect()
  ops.prof_enable(True)
  for _ in range(a.steps):
    step()
  torch.cuda.synchronize()
  ops.prof_enable(False)
  rows = ops.prof_collect_launches()
  per = collections.OrderedDict()      # (tag, kind) -> [ms, launches]
  for kind, tag, ms in rows:
    if not kind.startswith('conv'):
      continue
    e = per.setdefault((tag, kind), [0.0, 0])
    e[0] += ms
    e[1] += 1
  shapes = collections.OrderedDict()
  for (tag, kind), (ms, n) in per.items():
    s = shapes.setdefault(tag, {})
    s[kind] = (ms / a.steps * 1e3, n / a.steps)

  def load(path):
    if not path or not os.path.exists(path):
      return {}
    with open(path) as fh:
      rep = json.load(fh)
    out = {}
    for c in rep['convs']:
      key = tuple(c['shape'])
      e = out.setdefault(key, [0.0, 0.0, 0])
      e[0] += c['ms_fwd'] * 1e3
      e[1] += c['ms_bwd'] * 1e3
      e[2] += 1
    return out
  warm, cold = load(a.alone), load(a.cold)
  print('%-30s %3s | %9s %9s %9s | %9s %9s %9s | %6s' % ('shape (h, w, cin, cout, k, s)', 'n', 'fwd step', 'alone', 'cold',
                                                        'bwd step', 'alone', 'cold', 'launch'))
  tot = collections.Counter()
  table = []
  for tag, s in shapes.items():
    f_us = sum(v[0] for k, v in s.items() if k == 'conv_fwd')
    b_us = sum(v[0] for k, v in s.items() if k != 'conv_fwd')
    n_l = sum(v[1] for v in s.values())
    key = tuple(tag)
    w, c = warm.get(key), cold.get(key)
    n_layers = w[2] if w else (c[2] if c else 0)
    row = dict(shape=list(tag), layers=n_layers, fwd_step_us=f_us, bwd_step_us=b_us, launches_per_step=n_l,
               fwd_alone_us=w[0] if w else None, bwd_alone_us=w[1] if w else None,
               fwd_cold_us=c[0] if c else None, bwd_cold_us=c[1] if c else None)
    table.append(row)
    fmt = lambda v: ('%9.1f' % v) if v is not None else '%9s' % '-'
    print('%-30s %3d | %s %s %s | %s %s %s | %6.1f' % (tuple(tag), n_layers, fmt(f_us), fmt(row['fwd_alone_us']), fmt(row['fwd_cold_us']),
                                                       fmt(b_us), fmt(row['bwd_alone_us']), fmt(row['bwd_cold_us']), n_l))
    for k in ('fwd_step_us', 'bwd_step_us', 'fwd_alone_us', 'bwd_alone_us', 'fwd_cold_us', 'bwd_cold_us'):
      if row[k] is not None:
        tot[k] += row[k]
  print('TOTAL us per step: in the step fwd %.0f + bwd %.0f = %.0f | alone (warm) %.0f + %.0f = %.0f | alone (cold) %.0f + %.0f = %.0f' % (
      tot['fwd_step_us'], tot['bwd_step_us'], tot['fwd_step_us'] + tot['bwd_step_us'],
      tot['fwd_alone_us'], tot['bwd_alone_us'], tot['fwd_alone_us'] + tot['bwd_alone_us'],
      tot['fwd_cold_us'], tot['bwd_cold_us'], tot['fwd_cold_us'] + tot['bwd_cold_us']))
  os.makedirs(os.path.dirname(a.out), exist_ok=True)
  with open(a.out, 'w') as fh:
    json.dump(dict(batch=a.batch, steps=a.steps, rows=table, totals=dict(tot)), fh, indent=1)


if __name__ == '__main__':
  main()
