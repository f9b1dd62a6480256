#!/usr/bin/env python3
"""K1 tile sweep (development tool): the 8-wave ping-pong body (convpp.hpp) against the igemm body on the ResNet-50
long-reduction layers, forward and dgrad, at one or more batch sizes.

For every layer x pass x variant (0 = igemm body, 1 = 256x256, 2 = 128x256, 3 = 256x128, 4 = 512x128) it checks the
output against the igemm body's (both are bf16 roundings of fp32 sums in different orders: elements may differ by one
bf16 ulp; anything larger is reported as BAD), then times the kernel alone with events.  Also prints the dense bf16 MFMA
rate of the box (rigl_probe_mfma_bf16).  Writes gpurun_out/pp_sweep.json.

  python tools/pp_sweep.py --batch 128 512 --iters 20
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rigl_amd import ops  # noqa: E402

DEV = 'cuda:0'
# name, H, W, Cin, Cout, k, stride  (ResNet-50 v1.5 layers with >= 8 K-tiles of 64 in the forward or the dgrad GEMM)
LAYERS = [
    ('g3_c2_3x3_256', 14, 14, 256, 256, 3, 1),
    ('g3_c1_1024_256', 14, 14, 1024, 256, 1, 1),
    ('g3_c3_256_1024', 14, 14, 256, 1024, 1, 1),
    ('g4_c2_3x3_512', 7, 7, 512, 512, 3, 1),
    ('g4_c1_2048_512', 7, 7, 2048, 512, 1, 1),
    ('g4_c3_512_2048', 7, 7, 512, 2048, 1, 1),
    ('g2_c2_3x3_128', 28, 28, 128, 128, 3, 1),
    ('g2_c1_512_128', 28, 28, 512, 128, 1, 1),
    ('g2_c3_128_512', 28, 28, 128, 512, 1, 1),
    ('g3_b0_c1_512_256', 28, 28, 512, 256, 1, 1),
    ('g4_b0_c1_1024_512', 14, 14, 1024, 512, 1, 1),
    ('g3_b0_c2_3x3s2', 28, 28, 256, 256, 3, 2),
    ('g4_b0_c2_3x3s2', 14, 14, 512, 512, 3, 2),
    ('g3_proj_s2', 28, 28, 512, 1024, 1, 2),
    ('g4_proj_s2', 14, 14, 1024, 2048, 1, 2),
    ('g1_c2_3x3_64', 56, 56, 64, 64, 3, 1),
]
NAMES = {0: 'igemm', 1: '256x256', 2: '128x256', 3: '256x128', 4: '512x128'}


def timeit(fn, iters, warmup=3):
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


def compare(a, b):
  """(fraction of differing elements, max |a-b| / (|b| + tiny) over differing ones)"""
  af, bf = a.float(), b.float()
  ne = af != bf
  n = int(ne.sum())
  if n == 0:
    return 0.0, 0.0
  rel = ((af - bf).abs() / (bf.abs() + 1e-3))[ne].max().item()
  return n / af.numel(), rel


def bwd_rows(a, rep, name, B, H, W, Cin, Cout, k, s, pt, pl, Ho, Wo, macs, x, dy, hwio, add):
  """The one-call backward (pp_bwd 0 / 1 / 2) and the stand-alone weight gradient (pp_wgrad 0 / 1): dW against the
  igemm / tr kernels' (fp32 reassociation: relative to max |dW|), dX bit-equal to the stand-alone dgrad of the same
  setting, run-to-run bit determinism of dW; then the timings (backward time includes the split-K reduce launch)."""
  n = k * k * Cin * Cout
  has_dx = Cin % 8 == 0
  for ps, key, vals in (('bwd', 'pp_bwd', ((0, 0), (1, 0), (2, 0))), ('wgrad', 'pp_wgrad', ((0, 0), (1, 0)))):
    if ps not in a.passes:
      continue
    line = '%-20s B%-4d %-5s' % (name, B, ps)
    ref_dw = None
    for v, wk in vals:
      t = 128 if wk else 256
      if v and (Cin % t or Cout % t):
        continue
      ops.tune_set('pp_fwd', 0); ops.tune_set('pp_dgrad', -1); ops.tune_set('pp_bwd', 0); ops.tune_set('pp_wgrad', 0)
      ops.tune_set(key, v)
      d = ops.conv_desc(B, H, W, Cin, Cout, k, k, s, pt, pl, Ho, Wo)
      try:
        dw = torch.empty(n, dtype=torch.float32, device=DEV)
        if ps == 'bwd':
          dx = ops.conv_bwd(d, x, dy, hwio, dw, need_dx=has_dx, addend=add if has_dx else None)
          run = lambda: (ops.conv_bwd(d, x, dy, hwio, dw, need_dx=has_dx, addend=add if has_dx else None))
        else:
          ops.conv_wgrad(d, x, dy, dw)
          dx = None
          run = lambda: ops.conv_wgrad(d, x, dy, dw)
        torch.cuda.synchronize()
        bad = ''
        if ref_dw is None:
          ref_dw = dw.clone()
        else:
          e = (dw - ref_dw).abs().max().item() / (ref_dw.abs().max().item() + 1e-12)
          if not e < 2e-5:
            bad += ' BADDW(%.3g)' % e
        if dx is not None:
          dx2 = ops.conv_dgrad(d, dy, hwio, addend=add)
          if not torch.equal(dx.view(torch.int16), dx2.view(torch.int16)):
            bad += ' BADDX(%d)' % int((dx.float() != dx2.float()).sum())
        dw2 = torch.empty_like(dw)
        if ps == 'bwd':
          ops.conv_bwd(d, x, dy, hwio, dw2, need_dx=has_dx, addend=add if has_dx else None)
        else:
          ops.conv_wgrad(d, x, dy, dw2)
        if not torch.equal(dw.view(torch.int32), dw2.view(torch.int32)):
          bad += ' NONDET'
        ms = timeit(run, a.iters)
        tf = (4 if ps == 'bwd' and has_dx else 2) * macs / (ms * 1e-3) / 1e12
        rep['rows'].append(dict(layer=name, batch=B, kind=ps, variant=v, wk=wk, ms=ms, tflops=tf, bad=bad))
        line += ' | %s=%d%s %6.1f us %5.0f TF%s' % (key, v, '/wk' if wk else '', ms * 1e3, tf, bad)
      except Exception as ex:  # pylint: disable=broad-except
        line += ' | %s=%d FAILED %s' % (key, v, repr(ex)[:80])
    print(line, flush=True)
  ops.tune_set('pp_bwd', -1); ops.tune_set('pp_wgrad', -1); ops.tune_set('pp_fwd', -1)


DIMS = {1: (256, 256), 2: (128, 256), 3: (256, 128), 4: (512, 128)}


def legal(v, ps, B, H, W, Cin, Cout, s, Ho, Wo):
  """Mirror of pp_legal (convpp.hpp): where the library would silently keep the igemm body."""
  bm, bn = DIMS[v]
  M, N, K = (B * Ho * Wo, Cout, Cin) if ps == 'fwd' else (B * H * W, Cin, Cout)
  if ps == 'dgrad' and s > 2:
    return False
  return K % 64 == 0 and N % bn == 0 and M >= bm


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, nargs='+', default=[128])
  ap.add_argument('--iters', type=int, default=20)
  ap.add_argument('--variants', type=int, nargs='+', default=[0, 1, 2, 3, 4])
  ap.add_argument('--layers', nargs='*', default=None)
  ap.add_argument('--passes', nargs='+', default=['fwd', 'dgrad'])
  ap.add_argument('--ph', type=int, nargs='+', default=[0], help='pp_ph settings to sweep: 0 = built-in phases, 4 = four phases')
  ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'pp_sweep.json'))
  ap.add_argument('--knob', nargs='*', default=[], help='extra knobs for the whole run, key=value')
  a = ap.parse_args()
  for kv in a.knob:
    k_, v_ = kv.split('=')
    ops.tune_set(k_, int(v_))
  rep = dict(mfma_peak_tflops=ops.mfma_peak_probe(DEV), rows=[])
  print('MFMA bf16 dense rate of this box: %.0f TFLOP/s' % rep['mfma_peak_tflops'], flush=True)
  for B in a.batch:
    for (name, H, W, Cin, Cout, k, s) in LAYERS:
      if a.layers and name not in a.layers:
        continue
      p = (k - 1) // 2
      Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
      macs = B * Ho * Wo * Cout * k * k * Cin
      g = torch.Generator(device=DEV).manual_seed(1234)
      x = torch.randn(B, H, W, Cin, generator=g, device=DEV).to(torch.bfloat16)
      dy = torch.randn(B, Ho, Wo, Cout, generator=g, device=DEV).to(torch.bfloat16)
      w = torch.randn(k, k, Cin, Cout, generator=g, device=DEV) * (2.0 / (k * k * Cin)) ** 0.5
      add = torch.randn(B, H, W, Cin, generator=g, device=DEV).to(torch.bfloat16)
      n = w.numel()
      hwio = torch.empty(n, dtype=torch.bfloat16, device=DEV)
      ohwi = torch.empty(n, dtype=torch.bfloat16, device=DEV)
      ops.pack_weights(w.reshape(-1).contiguous(), None, k * k * Cin, Cout, hwio, ohwi)
      pt = pl = p
      ref = {}
      if 'bwd' in a.passes or 'wgrad' in a.passes:
        bwd_rows(a, rep, name, B, H, W, Cin, Cout, k, s, pt, pl, Ho, Wo, macs, x, dy, hwio, add)
      for ps in [q for q in a.passes if q in ('fwd', 'dgrad')]:
        line = '%-20s B%-4d %-5s' % (name, B, ps)
        for v, ph in [(v, ph) for v in a.variants for ph in (a.ph if v else [0])]:
          if v and not legal(v, ps, B, H, W, Cin, Cout, s, Ho, Wo):
            continue
          ops.tune_set('pp_ph', ph)
          ops.tune_set('pp_fwd', v)
          ops.tune_set('pp_dgrad', v)
          d = ops.conv_desc(B, H, W, Cin, Cout, k, k, s, pt, pl, Ho, Wo)
          try:
            if ps == 'fwd':
              y, part = ops.conv_fwd(d, x, ohwi, stats=True)
              out, st = y, part.double().sum(0)
              run = lambda: ops.conv_fwd(d, x, ohwi, stats=True)
            else:
              out = ops.conv_dgrad(d, dy, hwio, addend=add)
              st = None
              run = lambda: ops.conv_dgrad(d, dy, hwio, addend=add)
            torch.cuda.synchronize()
            bad = ''
            if v == 0:
              ref[ps] = (out.clone(), st)
            elif ps in ref:
              frac, rel = compare(out, ref[ps][0])
              if rel > 2.0 ** -6 or frac > 0.2:
                bad = ' BAD(frac %.3g rel %.3g)' % (frac, rel)
              if st is not None and ref[ps][1] is not None:
                den = ref[ps][1].abs().max().item() + 1e-6
                e = ((st - ref[ps][1]).abs().max().item()) / den
                if e > 1e-3:
                  bad += ' BADSTATS(%.3g)' % e
            ms = timeit(run, a.iters)
            tf = 2 * macs / (ms * 1e-3) / 1e12
            rep['rows'].append(dict(layer=name, batch=B, kind=ps, variant=v, ph=ph, ms=ms, tflops=tf, bad=bad))
            line += ' | %s%s %6.1f us %5.0f TF%s' % (NAMES[v], '/4ph' if ph == 4 else '', ms * 1e3, tf, bad)
          except Exception as ex:  # pylint: disable=broad-except
            line += ' | %s FAILED %s' % (NAMES[v], repr(ex)[:60])
        print(line, flush=True)
      del x, dy, w, add, hwio, ohwi, ref
  ops.tune_set('pp_fwd', -1)
  ops.tune_set('pp_dgrad', -1)
  ops.tune_set('pp_ph', 0)
  os.makedirs(os.path.dirname(a.out), exist_ok=True)
  with open(a.out, 'w') as f:
    json.dump(rep, f, indent=1)


if __name__ == '__main__':
  main()
