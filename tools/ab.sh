#!/bin/bash
# dev helper: run bench.py under several env settings on the same box, one summary line each
# usage: tools/ab.sh "label1:ENV=1 ENV2=x" "label2:" ...
for spec in "$@"; do
  label="${spec%%:*}"; envs="${spec#*:}"
  env $envs timeout 240 python bench.py --steps ${AB_STEPS:-60} --warmup ${AB_WARMUP:-15} --no-cpu-baseline ${AB_ARGS:-} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$label', round(d['value']), round(d['ms_per_step'],3), round(r['frac'],4), {k:round(v,3) for k,v in r['by_kind_ms_per_step'].items() if v>0.1})"
done
