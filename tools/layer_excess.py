#!/usr/bin/env python3
"""Per layer shape: K1 time inside the ResNet-50 step (tools/instep_table.py's JSON) against the shape's own bound,
max(flops / 2.5 PFLOP/s, algorithmic bytes / 6.3 TB/s) per pass -- forward: x + W + y; backward: (dY + W + dX) + (x + dY +
fp32 dW); a strided 1x1 conv reads a quarter of x; the fused gradient accumulation of a block input is NOT in the bound
(it is real traffic the layer carries for the graph).  Sorted by the time over the bound.  6.3 TB/s is what the
streaming kernels of this repo reach on HBM3E; 8 TB/s is the spec figure bench.py's layerwise_bound uses."""
import argparse
import json


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('instep_json')
  ap.add_argument('--batch', type=int, default=128)
  ap.add_argument('--tbs', type=float, default=6.3)
  a = ap.parse_args()
  d = json.load(open(a.instep_json))
  n_img, rows, bound_total, step_total = a.batch, [], 0.0, 0.0
  for r in d['rows']:
    h, w, cin, cout, k, s = r['shape']
    n = r['layers']
    ho, wo = (h + s - 1) // s, (w + s - 1) // s
    m, kk = n_img * ho * wo, k * k * cin
    flops = 2.0 * m * kk * cout
    x, y, wb = n_img * h * w * cin * 2.0, m * cout * 2.0, kk * cout * 2.0
    xr = x / 4 if (k == 1 and s == 2) else x
    f_b = max(flops / 2.5e15, (xr + y + wb) / (a.tbs * 1e12)) * 1e6
    b_bytes = (y + wb + x) + (xr + y + kk * cout * 4.0)
    b_flops = 2 * flops
    if cin == 3:                      # the stem has no dX
      b_bytes, b_flops = xr + y + kk * cout * 4.0, flops
    b_b = max(b_flops / 2.5e15, b_bytes / (a.tbs * 1e12)) * 1e6
    fs, bs = r['fwd_step_us'] / n, r['bwd_step_us'] / n
    rows.append((n * (fs - f_b) + n * (bs - b_b), tuple(r['shape']), n, fs, f_b, bs, b_b))
    bound_total += n * (f_b + b_b)
    step_total += n * (fs + bs)
  print('shape (h, w, cin, cout, k, s)     n | over the bound | fwd us (bound) | bwd us (bound)   per layer')
  for e in sorted(rows, reverse=True):
    print('%-30s x%d | %7.0f us     | %5.1f (%5.1f)  | %6.1f (%5.1f)' % (str(e[1]), e[2], e[0], e[3], e[4], e[5], e[6]))
  print('K1 in the step %.0f us per step; sum of the bounds %.0f us (%.2f of it)' % (step_total, bound_total, bound_total / step_total))


if __name__ == '__main__':
  main()
