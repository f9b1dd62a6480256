#!/bin/bash
# bench.py after the round-2 rewrite: contract tests, the four workloads, default run
cd "$(dirname "$0")/.."
O=gpurun_out/r2h; mkdir -p $O
timeout 1200 python -m pytest tests/test_dp_smoke_gpu.py -x -q -m gpu 2>&1 | tail -15 | tee $O/log.txt
for w in resnet50 resnet50_erk99 mobilenet_v1 wrn22; do
  timeout 300 python bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_$w.json 2>$O/err_$w.txt; echo "$w rc=$?" | tee -a $O/log.txt
  python - $O/bench_$w.json <<'PY' | tee -a $O/log.txt
import json,sys
try:
  d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); r=d['roofline']
  print(' ', round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms  frac', round(r['frac'],4), 'gflop/img', round(r['algorithmic_gflop_per_image'],3),
        {k:(round(v['achieved']), round(v.get('ms_per_update', v.get('ms_per_step',0)),4)) for k,v in r['hbm_kernels'].items()})
except Exception as e: print('parse failed', e); print(open(sys.argv[1].replace('bench_','err_').replace('.json','.txt')).read()[-1500:])
PY
done
timeout 600 python bench.py > $O/bench_default.json 2>$O/err_default.txt; echo "default rc=$?" | tee -a $O/log.txt
tail -1 $O/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])" | tee -a $O/log.txt
