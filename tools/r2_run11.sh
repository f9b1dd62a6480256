#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2k; mkdir -p $O; : > $O/log.txt
for cfg in "wrn22 " "wrn22 --graph" "mobilenet_v1 " "mobilenet_v1 --graph" "resnet50 " "resnet50 --graph"; do
  timeout 300 python bench.py --workload $cfg --steps 120 --warmup 10 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "
import json,sys
try:
  d=json.loads(sys.stdin.read()); r=d['roofline']
  print('$cfg', round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms frac', round(r['frac'],4), d['config']['execution'], d['config']['mask_updates_in_timed_region'])
except Exception as e: print('$cfg FAILED', e); print(open('$O/err.txt').read()[-2000:])" | tee -a $O/log.txt
done
