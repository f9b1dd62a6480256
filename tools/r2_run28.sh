#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2dd
cd $R
{
timeout 600 python -m pytest tests/test_dp_smoke_gpu.py -q -m gpu -x 2>&1 | tail -4
RIGL_BENCH_FORCE_SYNC=1 timeout 200 python bench.py --steps 60 --warmup 15 --no-cpu-baseline > $R/gpurun_out/r2dd/out.txt 2> $R/gpurun_out/r2dd/err.txt
echo rc=$? lines=$(wc -l < $R/gpurun_out/r2dd/out.txt)
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2dd/out.txt').read().strip())
a=d['allreduce']
print(round(d['value']), round(d['ms_per_step'],3), d['roofline']['frac'])
print({k:a[k] for k in a if k!='in_step'})
print(a['in_step'])
PY
timeout 200 python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('plain', round(d['value']), round(d['ms_per_step'],3))"
} > $R/gpurun_out/r2dd/log.txt 2>&1
cat $R/gpurun_out/r2dd/log.txt
