#!/usr/bin/env python3
"""Per-kernel summary of one rocprofv3 PMC pass (csv): every collected counter per call, plus ratios to SQ_WAVE_CYCLES.
usage: pmc_kernel_summary.py <..._counter_collection.csv> [name-filter]"""
import collections
import csv
import sys


def main():
  flt = sys.argv[2] if len(sys.argv) > 2 else 'rigl::k1'
  agg = collections.defaultdict(lambda: collections.defaultdict(float))
  calls = collections.defaultdict(set)
  dur = collections.defaultdict(float)
  for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('void ', '')
    if flt not in n:
      continue
    n = n.split('(')[0]
    agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    did = r.get('Dispatch_Id', r.get('Correlation_Id'))
    if did not in calls[n]:
      calls[n].add(did)
      dur[n] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
  for n, c in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    k = len(calls[n])
    wc = c.get('SQ_WAVE_CYCLES', 0.0)
    print('%s  calls %d  us/call %.1f' % (n[:70], k, dur[n] / k / 1e3))
    for name, v in sorted(c.items()):
      extra = ''
      if wc and name.startswith('SQ_') and name != 'SQ_WAVE_CYCLES':
        extra = '  (%.1f%% of SQ_WAVE_CYCLES)' % (100 * v / wc)
      print('    %-28s %14.0f per call%s' % (name, v / k, extra))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
      print('    mfma busy vs 2.4 GHz x 1024 SIMDs: %.1f%%' % (100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (dur[n] * 2.4 * 1024)))
    if 'SQ_LDS_IDX_ACTIVE' in c:
      print('    LDS array busy: %.1f%% of kernel time x 256 CUs (IDX_ACTIVE / (t x 2.4 GHz x 256)); conflicts %.1f%% of it' % (
          100 * c['SQ_LDS_IDX_ACTIVE'] / (dur[n] * 2.4 * 256), 100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c['SQ_LDS_IDX_ACTIVE'], 1)))


if __name__ == '__main__':
  main()
