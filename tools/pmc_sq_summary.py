#!/usr/bin/env python3
"""MFMA-busy and wave-state summary of the K1 kernels from one rocprofv3 PMC pass:

  cd /tmp && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY \
      SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
      --output-format csv -d gpurun_out/pmc_sq -o pmc -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-prof
  python tools/pmc_sq_summary.py gpurun_out/pmc_sq/pmc_counter_collection.csv > profiles/r1/pmc_sq_k1.txt

mfma%pk = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x 2.4 GHz x 1024 SIMDs): the share of the chip's MFMA
issue capacity at the 2.4 GHz the 2.5 PFLOP/s peak assumes (kernel time from the dispatch timestamps; one
32x32x16 bf16 MFMA = 32 busy cycles).  wait / winst / active: SQ_WAIT_ANY (parked on s_waitcnt / barrier),
SQ_WAIT_INST_ANY (issue stalls) and SQ_ACTIVE_INST_ANY as fractions of SQ_WAVE_CYCLES.
"""
import collections
import csv
import sys


def main():
  agg = collections.defaultdict(lambda: collections.defaultdict(float))
  dur = collections.defaultdict(float)
  cnt = collections.Counter()
  for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if 'rigl::k1' not in n:
      continue
    agg[n][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
      cnt[n] += 1
      dur[n] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
  print('%-52s %6s %8s %8s %7s %7s %8s %9s' % ('kernel', 'calls', 'us/call', 'mfma%pk', 'wait%', 'winst%', 'active%', 'ldsconf%'))
  tm = td = 0.0
  for n, c in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    d, wc = dur[n], c['SQ_WAVE_CYCLES']
    tm += c['SQ_VALU_MFMA_BUSY_CYCLES']
    td += d
    print('%-52s %6d %8.1f %7.1f%% %6.1f%% %6.1f%% %7.1f%% %8.1f%%' % (
        n[:52], cnt[n], d / cnt[n] / 1e3, 100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (d * 2.4 * 1024),
        100 * c['SQ_WAIT_ANY'] / wc, 100 * c['SQ_WAIT_INST_ANY'] / wc, 100 * c['SQ_ACTIVE_INST_ANY'] / wc,
        100 * c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1.0)))
  print('K1 overall MFMA busy vs 2.4 GHz x 1024 SIMDs: %.1f%%' % (100 * tm / (td * 2.4 * 1024)))


if __name__ == '__main__':
  main()
