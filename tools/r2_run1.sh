#!/bin/bash
# GPU run 1 of round 2: tile196 correctness + per-layer A/B + step A/B.  Outputs under gpurun_out/r2a/.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
echo "== k1_check t196 forced" | tee $O/log.txt
RIGL_T196=2 timeout 900 python tests/k1_check.py --set t196 > $O/k1_t196_forced.txt 2>&1; echo "rc=$?" | tee -a $O/log.txt; tail -3 $O/k1_t196_forced.txt | tee -a $O/log.txt
echo "== k1_check resnet50 b128 default" | tee -a $O/log.txt
timeout 1500 python tests/k1_check.py --set resnet50 --batch 128 > $O/k1_r50_default.txt 2>&1; echo "rc=$?" | tee -a $O/log.txt; tail -3 $O/k1_r50_default.txt | tee -a $O/log.txt
echo "== k1_check resnet50 b128 forced t196 + separate bwd" | tee -a $O/log.txt
RIGL_T196=2 RIGL_T196_BWD=1 timeout 1500 python tests/k1_check.py --set resnet50 --batch 128 > $O/k1_r50_forced.txt 2>&1; echo "rc=$?" | tee -a $O/log.txt; tail -3 $O/k1_r50_forced.txt | tee -a $O/log.txt
for m in 0 2; do
  echo "== bench_kernels RIGL_T196=$m" | tee -a $O/log.txt
  RIGL_T196=$m timeout 600 python tools/bench_kernels.py --out $O/bench_kernels_t196_$m.json > $O/bench_kernels_t196_$m.txt 2>&1; echo "rc=$?" | tee -a $O/log.txt
  tail -4 $O/bench_kernels_t196_$m.txt | tee -a $O/log.txt
done
for cfg in "RIGL_T196=0" "RIGL_T196=1" "RIGL_T196=2" "RIGL_T196=2 RIGL_T196_BWD=1" "RIGL_T196=1 RIGL_T196_BWD=1"; do
  echo "== bench.py $cfg" | tee -a $O/log.txt
  env $cfg timeout 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --prof-every 5 > "$O/bench_$(echo $cfg | tr ' =' '__').json" 2>$O/bench_err.txt; echo "rc=$?" | tee -a $O/log.txt
  python - "$O/bench_$(echo $cfg | tr ' =' '__').json" <<'PY' | tee -a $O/log.txt
import json,sys
try:
  d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
  r=d['roofline']
  print('value %.0f img/s  ms/step %.3f  conv_ms %.3f frac %.4f  by_kind %s' % (d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['frac'], {k:round(v,3) for k,v in r['by_kind_ms_per_step'].items()}))
except Exception as e:
  print('parse failed', e)
PY
done
echo "== pytest gpu (existing K1 tests)" | tee -a $O/log.txt
timeout 1200 python -m pytest tests/test_k3_k1_gpu.py tests/test_fullsize_properties_gpu.py -x -q -m gpu > $O/pytest_k1.txt 2>&1; echo "rc=$?" | tee -a $O/log.txt; tail -5 $O/pytest_k1.txt | tee -a $O/log.txt
