#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2j; mkdir -p $O
timeout 600 python -m pytest tests/test_k3_k1_gpu.py -x -q -m gpu -k depthwise 2>&1 | tail -3 | tee $O/log.txt
for cfg in "RIGL_DW_PARTS=512" "RIGL_DW_PARTS=1024" "RIGL_DW_PARTS=256"; do
  env $cfg timeout 300 python bench.py --workload mobilenet_v1 --steps 40 --warmup 10 --no-cpu-baseline 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$cfg', round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms frac', round(r['frac'],4), {k:(round(v['achieved']), round(v.get('ms_per_update', v.get('ms_per_step',0)),4)) for k,v in r['hbm_kernels'].items()})" | tee -a $O/log.txt
done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -o mb -- python $OLDPWD/bench.py --workload mobilenet_v1 --steps 20 --warmup 5 --no-cpu-baseline --no-prof > $OLDPWD/$O/prof_run.txt 2>&1
cd $OLDPWD
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/mobilenet_kernel_stats.csv; find $O/prof -name "*.csv" ! -name "*stats*" -delete
grep kdw $O/mobilenet_kernel_stats.csv | cut -c1-160 | tee -a $O/log.txt
