#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2q; mkdir -p $O; : > $O/log.txt
for cfg in "RIGL_DW_ROLL=0" "RIGL_DW_ROLL_MIN=150000" "RIGL_DW_ROLL_MIN=100000" "RIGL_DW_ROLL_MIN=50000" "RIGL_DW_ROLL=0" "RIGL_DW_ROLL_MIN=150000"; do
  env $cfg timeout 300 python bench.py --workload mobilenet_v1 --steps 60 --warmup 15 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$cfg', round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms', {k:(round(v['achieved']), round(v.get('ms_per_update', v.get('ms_per_step',0)),4)) for k,v in r['hbm_kernels'].items() if k=='depthwise_conv'})" | tee -a $O/log.txt
done
R=$PWD; cd /tmp; export TMPDIR=/tmp
RIGL_DW_ROLL_MIN=100000 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o mb -- python $R/bench.py --workload mobilenet_v1 --steps 20 --warmup 5 --no-cpu-baseline --no-prof > $R/$O/prof_run.txt 2>&1
cd $R; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/mobilenet_kernel_stats.csv; rm -rf $O/prof
grep kdw $O/mobilenet_kernel_stats.csv | cut -c1-170 | tee -a $O/log.txt
