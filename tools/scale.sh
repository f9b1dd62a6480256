#!/bin/bash
# The 1 -> 8 GPU scaling set of one node, with nobody in the loop:  tools/scale.sh <tag>  writes
# profiles/<tag>/scale.jsonl (one bench.py line per run, the N > 1 lines carrying allreduce.exposed.comm_exposed_ms and
# masks_identical_across_ranks) and profiles/<tag>/scale_summary.txt.
#   1. python bench.py --gpus N for N in 1 2 4 8 (as many as the node has): weak scaling, per-GPU batch 128
#   2. a RIGL_DP_BUCKET_MB sweep (8 16 32 64 128) at the largest N: the bucket size of the gradient-arena all-reduce
# bench.py --gpus N launches itself under torch.distributed.run on 127.0.0.1 (one process per GPU, RCCL over xGMI).
# Reference: CrossShardOptimizer / cross_replica_sum (imagenet_train_eval.py:363-365, sparse_optimizers_base.py:472-473).
set -u
TAG=${1:-scale}
R="$(cd "$(dirname "$0")/.." && pwd)"
O=$R/profiles/$TAG; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
NG=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
echo "GPUs on this node: $NG" | tee $O/scale_summary.txt
: > $O/scale.jsonl
MAXN=1
for N in 1 2 4 8; do
  [ "$N" -le "$NG" ] || continue
  MAXN=$N
  timeout 900 python bench.py --gpus $N --steps ${SCALE_STEPS:-100} --warmup ${SCALE_WARMUP:-20} --no-cpu-baseline 2>>$O/scale_err.txt | tail -1 >> $O/scale.jsonl
  echo "gpus=$N rc=$?" | tee -a $O/scale_summary.txt
done
if [ "$MAXN" -gt 1 ]; then
  for MB in 8 16 32 64 128; do
    RIGL_DP_BUCKET_MB=$MB timeout 900 python bench.py --gpus $MAXN --steps ${SCALE_STEPS:-100} --warmup ${SCALE_WARMUP:-20} --no-cpu-baseline 2>>$O/scale_err.txt \
      | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); d['bucket_mb']=$MB; print(json.dumps(d))" >> $O/scale.jsonl
    echo "gpus=$MAXN bucket_mb=$MB rc=$?" | tee -a $O/scale_summary.txt
  done
fi
python - "$O/scale.jsonl" <<'P' | tee -a $O/scale_summary.txt
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip().startswith('{')]
base = next((r['value'] for r in rows if r['n_gpus'] == 1 and 'bucket_mb' not in r), None)
for r in rows:
  ar = r.get('allreduce') or {}
  ex = (ar.get('exposed') or {}).get('comm_exposed_ms')
  print('n_gpus %d%s: %8.0f images/s, %.3f ms/step, x%.2f of one GPU, exposed communication %s ms, masks identical across ranks: %s' % (
      r['n_gpus'], ' bucket %d MB' % r['bucket_mb'] if 'bucket_mb' in r else '', r['value'], r['ms_per_step'],
      r['value'] / base if base else float('nan'), 'n/a' if ex is None else '%.3f' % ex, ar.get('masks_identical_across_ranks')))
P
