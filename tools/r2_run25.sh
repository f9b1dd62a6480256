#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2prof
cd $R
for i in 1 2 3; do timeout 300 python bench.py 2>/dev/null | tail -1 >> gpurun_out/r2prof/bench_n1_runs.jsonl; done
python - <<'PY'
import json
for l in open('gpurun_out/r2prof/bench_n1_runs.jsonl'):
    d=json.loads(l); r=d['roofline']
    print(round(d['value']), round(d['ms_per_step'],3), round(r['frac'],4), round(r['conv_ms_per_step'],3), r['traffic_from_this_build'], round(d['cpu_baseline']['value'],1))
PY
