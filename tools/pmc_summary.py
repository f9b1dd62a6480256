#!/usr/bin/env python3
"""Summarise the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- they do not
fit one pass on gfx950) of `bench.py --steps 4 --warmup 2 --no-cpu-baseline
--no-prof` into per-kernel HBM traffic, and write the K1 per-launch figure
bench.py reports as roofline.traffic.

  cd /tmp && for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c \
      --output-format csv -d gpurun_out/pmc_$c -o pmc -- python bench.py ...; done
  python tools/pmc_summary.py gpurun_out profiles/r1

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KB;
on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) streaming reads
(128-B requests tallied at 64 B) -> doubled here; WRITE_SIZE is used as is
(checked against the known output sizes of the 128x64-tile forward launches:
73.9 MB measured vs 73.4 MB of outputs per launch).
"""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lib_sha16():
  """The build the counters were collected on (bench.py compares it with the library it runs)."""
  try:
    with open(os.environ.get('RIGL_HIP_LIB') or os.path.join(ROOT, 'rigl_amd', 'lib', 'librigl_hip.so'), 'rb') as fh:
      return hashlib.sha256(fh.read()).hexdigest()[:16]
  except OSError:
    return None


def load(path):
  d = collections.defaultdict(lambda: [0, 0.0])
  for r in csv.DictReader(open(path)):
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    d[name][0] += 1
    d[name][1] += float(r['Counter_Value'])
  return d


def main():
  src, dst = sys.argv[1], sys.argv[2]
  f = load(os.path.join(src, 'pmc_FETCH_SIZE', 'pmc_counter_collection.csv'))
  w = load(os.path.join(src, 'pmc_WRITE_SIZE', 'pmc_counter_collection.csv'))
  rows = []
  tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
  for k in sorted(f, key=lambda k: -f[k][1]):
    n = f[k][0]
    rd = 2.0 * f[k][1] * 1024.0
    wr = w.get(k, [0, 0.0])[1] * 1024.0
    rows.append((k, n, rd / n, wr / n))
    fam = 'K1' if 'rigl::k1' in k else 'BN' if 'rigl::kbn' in k else 'other'
    tot[fam][0] += n
    tot[fam][1] += rd
    tot[fam][2] += wr
  with open(os.path.join(dst, 'pmc_hbm_traffic.csv'), 'w') as fh:
    fh.write('kernel,launches,read_bytes_per_launch(FETCH_SIZE*2),write_bytes_per_launch(WRITE_SIZE)\n')
    for r in rows:
      fh.write('"%s",%d,%.0f,%.0f\n' % r)
    for fam, (n, rd, wr) in tot.items():
      fh.write('"TOTAL %s",%d,%.0f,%.0f\n' % (fam, n, rd / n, wr / n))
  n, rd, wr = tot['K1']
  out = {'kernel': 'K1 (all rigl::k1 launches of 6 ResNet-50 steps, batch 128)', 'launches': n,
         'read_bytes_per_launch': rd / n, 'write_bytes_per_launch': wr / n,
         'bytes_per_launch': (rd + wr) / n, 'lib_sha16': lib_sha16(),
         'method': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950 '
                   'wide-read correction), KB -> bytes'}
  json.dump(out, open(os.path.join(dst, 'k1_traffic.json'), 'w'), indent=1)
  print(json.dumps(out))


if __name__ == '__main__':
  main()
