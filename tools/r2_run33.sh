#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2ii
cd $R
{
for mb in 32 16 8 32 16 8; do
RIGL_DP_BUCKET_MB=$mb RIGL_BENCH_FORCE_SYNC=1 timeout 200 python bench.py --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); a=d['allreduce']
rows=a['in_step']['buckets']
print('bucket $mb MB:', round(d['value']), round(d['ms_per_step'],3), 'buckets', a['buckets'], 'exposed', round(a['exposed']['comm_exposed_ms'],3), 'tail_wait', round(a['in_step']['tail_wait_ms'],3), 'last bucket MB', round(rows[-1]['bytes']/1e6,1), 'launch ms', [round(r['device_ms_after_first_bucket'],2) for r in rows])"
done
} > $R/gpurun_out/r2ii/log.txt 2>&1
cat $R/gpurun_out/r2ii/log.txt
