#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2g; mkdir -p $O
timeout 2400 python -m pytest tests/test_k1_parity_gpu.py tests/test_k3_k1_gpu.py tests/test_fullsize_properties_gpu.py -x -q -m gpu > $O/pytest_k1.txt 2>&1; echo "pytest rc=$?" | tee $O/log.txt; tail -4 $O/pytest_k1.txt | tee -a $O/log.txt
for cfg in "RIGL_T196=0 RIGL_C3=0" "RIGL_T196=1 RIGL_C3=1" "RIGL_T196=0 RIGL_C3=0" "RIGL_T196=1 RIGL_C3=1"; do
  env $cfg timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --prof-every 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$cfg', round(d['value']), round(d['ms_per_step'],3), round(r['frac'],4), {k:round(v,3) for k,v in r['by_kind_ms_per_step'].items() if v>0.1})" | tee -a $O/log.txt
done
