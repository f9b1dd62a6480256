#!/usr/bin/env python3
"""Per-kernel micro-benchmark on the ResNet-50 layer shapes (batch B per GPU).

Times each distinct conv shape (fwd / dgrad / wgrad) with HIP events and prints
achieved dense-equivalent TFLOP/s, plus K2 (whole-model prune/regrow) and K3
(whole-model masked momentum) in GB/s of ALGORITHMIC bytes.  Writes a JSON
report to gpurun_out/bench_kernels.json.  Development tool, not bench.py.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import ops  # noqa: E402
from tests.golden import layer_shapes  # noqa: E402


def resnet50_convs(batch):
  """(name, N,H,W,Cin,Cout,k,stride,pad,Ho,Wo) for the 53 convs + FC."""
  out = [('stem', batch, 224, 224, 3, 64, 7, 2, 3, 112, 112)]
  in_ch, hw = 64, 56
  for g, nb in enumerate([3, 4, 6, 3], start=1):
    f = 64 * 2**(g - 1)
    stride = 1 if g == 1 else 2
    ohw = hw // stride
    out.append(('g%d_proj' % g, batch, hw, hw, in_ch, 4 * f, 1, stride, 0, ohw, ohw))
    out.append(('g%d_b0_c1' % g, batch, hw, hw, in_ch, f, 1, 1, 0, hw, hw))
    out.append(('g%d_b0_c2' % g, batch, hw, hw, f, f, 3, stride, 1, ohw, ohw))
    out.append(('g%d_b0_c3' % g, batch, ohw, ohw, f, 4 * f, 1, 1, 0, ohw, ohw))
    for b in range(1, nb):
      out.append(('g%d_b%d_c1' % (g, b), batch, ohw, ohw, 4 * f, f, 1, 1, 0, ohw, ohw))
      out.append(('g%d_b%d_c2' % (g, b), batch, ohw, ohw, f, f, 3, 1, 1, ohw, ohw))
      out.append(('g%d_b%d_c3' % (g, b), batch, ohw, ohw, f, 4 * f, 1, 1, 0, ohw, ohw))
    in_ch, hw = 4 * f, ohw
  out.append(('fc', batch, 1, 1, 2048, 1000, 1, 1, 0, 1, 1))
  return out


def timeit(fn, iters=10, warmup=3):
  for _ in range(warmup):
    fn()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) / iters


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=128)
  ap.add_argument('--iters', type=int, default=10)
  ap.add_argument('--cold', action='store_true',
                  help='rotate every call through enough copies of the operand tensors that their combined footprint '
                       'exceeds 768 MB (3x the 256 MB Infinity Cache): the kernels then read their operands from HBM, as '
                       'they do inside the training step, instead of finding them cache-resident from the previous call')
  ap.add_argument('--no-k23', action='store_true', help='skip the K2 / K3 whole-model timings')
  ap.add_argument('--addend', action='store_true',
                  help="the one-call backward of every block's first conv (*_c1) also adds the shortcut gradient in its "
                       'dgrad epilogue, as it does in the training step (one more tensor of the size of dX read)')
  ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'bench_kernels.json'))
  a = ap.parse_args()
  dev = 'cuda:0'
  rep = dict(batch=a.batch, cold=bool(a.cold), convs=[], totals={})
  seen = {}
  tot = dict(fwd=0.0, dgrad=0.0, wgrad=0.0, flops_fwd=0.0, flops_dgrad=0.0)
  for (name, N, H, W, Cin, Cout, k, s, p, Ho, Wo) in resnet50_convs(a.batch):
    key = (H, W, Cin, Cout, k, s)
    with_add = a.addend and name.endswith('_c1')
    skey = key + (with_add,)
    macs = N * Ho * Wo * Cout * k * k * Cin
    if skey not in seen:
      try:
        d = ops.conv_desc(N, H, W, Cin, Cout, k, k, s, p, p, Ho, Wo)
        set_bytes = 2 * (N * H * W * Cin + N * Ho * Wo * Cout) * 2
        copies = max(2, -(-768 * (1 << 20) // set_bytes)) if a.cold else 1
        xs = [torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16) for _ in range(copies)]
        dys = [torch.randn(N, Ho, Wo, Cout, device=dev).to(torch.bfloat16) for _ in range(copies)]
        w = torch.randn(k * k * Cin * Cout, device=dev).to(torch.bfloat16)
        ys = [torch.empty(N, Ho, Wo, Cout, device=dev, dtype=torch.bfloat16) for _ in range(copies)]
        dxs = [torch.empty(N, H, W, Cin, device=dev, dtype=torch.bfloat16) for _ in range(copies)]
        dw = torch.empty(k * k * Cin * Cout, device=dev, dtype=torch.float32)
        turn = [0]

        def nxt():
          turn[0] = (turn[0] + 1) % copies
          return turn[0]
        iters = max(a.iters, copies) if a.cold else a.iters
        t_f = timeit(lambda: (lambda i: ops.conv_fwd(d, xs[i], w, ys[i]))(nxt()), iters)
        t_d = timeit(lambda: (lambda i: ops.conv_dgrad(d, dys[i], w, dxs[i]))(nxt()), iters) if Cin % 8 == 0 else 0.0
        t_w = timeit(lambda: (lambda i: ops.conv_wgrad(d, xs[i], dys[i], dw))(nxt()), iters)
        # dgrad + wgrad as the training step runs them: one launch (rigl_masked_conv2d_bwd)
        adds = [torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16) for _ in range(copies)] if with_add else None
        t_b = timeit(lambda: (lambda i: ops.conv_bwd(d, xs[i], dys[i], w, dw, need_dx=Cin % 8 == 0,
                                                     addend=adds[i] if with_add else None))(nxt()), iters)
        del adds
        seen[skey] = (t_f, t_d, t_w, t_b)
        del xs, dys, w, ys, dxs, dw
      except Exception as e:  # pylint: disable=broad-except
        seen[skey] = (float('nan'),) * 4
        print('FAILED', name, key, repr(e), flush=True)
    t_f, t_d, t_w, t_b = seen[skey]
    tf = lambda t: (2 * macs / (t * 1e-3) / 1e12) if t and t == t else 0.0
    rep['convs'].append(dict(name=name, shape=key, macs=macs, ms_fwd=t_f, ms_dgrad=t_d, ms_wgrad=t_w, ms_bwd=t_b,
                             tflops_fwd=tf(t_f), tflops_dgrad=tf(t_d), tflops_wgrad=tf(t_w)))
    print('%-10s %-28s fwd %7.3f ms %6.1f TF | dgrad %7.3f ms %6.1f TF | wgrad %7.3f ms %6.1f TF | bwd (one launch) %7.3f ms' % (
        name, key, t_f, tf(t_f), t_d, tf(t_d), t_w, tf(t_w), t_b), flush=True)
    tot['bwd'] = tot.get('bwd', 0.0) + t_b
    tot['fwd'] += t_f
    tot['dgrad'] += t_d
    tot['wgrad'] += t_w
    tot['flops_fwd'] += 2 * macs
    if name != 'stem':
      tot['flops_dgrad'] += 2 * macs
  conv_ms = tot['fwd'] + tot['dgrad'] + tot['wgrad']
  flops = 2 * tot['flops_fwd'] + tot['flops_dgrad']
  rep['totals'] = dict(ms_fwd=tot['fwd'], ms_dgrad=tot['dgrad'], ms_wgrad=tot['wgrad'], ms_conv=conv_ms,
                       tflops_conv=flops / (conv_ms * 1e-3) / 1e12,
                       img_per_s_conv_only=a.batch / (conv_ms * 1e-3))
  print('TOTAL conv: fwd %.2f ms dgrad %.2f ms wgrad %.2f ms = %.2f ms -> %.1f TFLOP/s, %.0f img/s (conv only)' % (
      tot['fwd'], tot['dgrad'], tot['wgrad'], conv_ms, rep['totals']['tflops_conv'],
      rep['totals']['img_per_s_conv_only']), flush=True)
  fused_ms = tot['fwd'] + tot.get('bwd', 0.0)
  rep['totals']['ms_bwd_one_launch'] = tot.get('bwd', 0.0)
  rep['totals']['tflops_conv_fused'] = flops / (fused_ms * 1e-3) / 1e12
  print('TOTAL with dgrad+wgrad in one launch: fwd %.2f ms + bwd %.2f ms = %.2f ms -> %.1f TFLOP/s' % (
      tot['fwd'], tot.get('bwd', 0.0), fused_ms, rep['totals']['tflops_conv_fused']), flush=True)

  if a.no_k23:
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, 'w') as f:
      json.dump(rep, f, indent=1)
    return
  # ---- K2 / K3 on the whole model ------------------------------------------
  shapes = list(layer_shapes.resnet50().values())
  layers = []
  for sh in shapes:
    n = int(np.prod(sh))
    layers.append(dict(w=torch.randn(n, device=dev) * 0.05, momentum=torch.zeros(n, device=dev),
                       dense_grad=torch.randn(n, device=dev) * 1e-3,
                       mask_bits=ops.mask_pack((torch.rand(n, device=dev) < 0.2).float())))
  n_tot = sum(l['w'].numel() for l in layers)
  t_k2 = timeit(lambda: ops.prune_regrow(layers, 0.3), 5, 2)
  rep['k2'] = dict(ms=t_k2, n=n_tot, gbps_algorithmic=8.25 * n_tot / (t_k2 * 1e-3) / 1e9)
  print('K2 prune/regrow whole ResNet-50: %.3f ms, %.1f GB/s algorithmic (8.25 B/weight)' % (
      t_k2, rep['k2']['gbps_algorithmic']), flush=True)
  flat_w = torch.randn(n_tot, device=dev)
  flat_m = torch.zeros(n_tot, device=dev)
  flat_g = torch.randn(n_tot, device=dev)
  flat_bits = ops.mask_pack((torch.rand(n_tot, device=dev) < 0.2).float())
  shadow = torch.empty(n_tot, device=dev, dtype=torch.bfloat16)
  t_k3 = timeit(lambda: ops.masked_sgd_momentum(flat_w, flat_g, 0.1, momentum=flat_m, mask_bits=flat_bits, mu=0.9,
                                                weight_decay=1e-4, nesterov=True, w_shadow=shadow), 20, 3)
  rep['k3'] = dict(ms=t_k3, n=n_tot, gbps_algorithmic=22.125 * n_tot / (t_k3 * 1e-3) / 1e9)
  print('K3 masked momentum whole ResNet-50: %.3f ms, %.1f GB/s algorithmic (22.1 B/weight incl. shadow)' % (
      t_k3, rep['k3']['gbps_algorithmic']), flush=True)
  os.makedirs(os.path.dirname(a.out), exist_ok=True)
  with open(a.out, 'w') as f:
    json.dump(rep, f, indent=1)


if __name__ == '__main__':
  main()
