#!/bin/bash
# The profile set of a round, all from one box and one build:  tools/profiles.sh <tag>  (e.g. r3) writes
# gpurun_out/<tag>prof/, which is then copied to profiles/<tag>/ and committed.  Replaces the per-round run scripts.
set -u
TAG=${1:-r3}
R="$(cd "$(dirname "$0")/.." && pwd)"
O=$R/gpurun_out/${TAG}prof; mkdir -p $O
cd $R
timeout 600 python bench.py > $O/bench_n1.json 2>$O/bench_err.txt; echo "bench rc=$?" | tee $O/log.txt
for w in resnet50_erk99 mobilenet_v1 wrn22; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>>$O/bench_err.txt; echo "$w rc=$?" | tee -a $O/log.txt
done
# (wrn22 replays a captured HIP graph by default since round 6; the eager line for comparison)
timeout 300 python bench.py --workload wrn22 --no-graph --no-cpu-baseline > $O/bench_wrn22_eager.json 2>>$O/bench_err.txt; echo "wrn22 eager rc=$?" | tee -a $O/log.txt
timeout 600 python tools/bench_kernels.py > $O/bench_kernels_per_layer.txt 2>&1; echo "bench_kernels rc=$?" | tee -a $O/log.txt
if [ "${SKIP_SWEEP:-0}" != "1" ]; then
timeout 600 python tools/pp_sweep.py --batch 128 --iters 20 --out $O/pp_sweep_b128.json > $O/pp_sweep_b128.txt 2>&1; echo "pp_sweep 128 rc=$?" | tee -a $O/log.txt
timeout 600 python tools/pp_sweep.py --batch 512 --iters 10 --layers g3_c2_3x3_256 g3_c1_1024_256 g3_c3_256_1024 g4_c2_3x3_512 g2_c2_3x3_128 --out $O/pp_sweep_b512.json > $O/pp_sweep_b512.txt 2>&1; echo "pp_sweep 512 rc=$?" | tee -a $O/log.txt
timeout 600 python tools/pp_sweep.py --batch 128 --iters 10 --passes bwd wgrad --out $O/pp_sweep_bwd_b128.json > $O/pp_sweep_bwd_b128.txt 2>&1; echo "pp_sweep bwd rc=$?" | tee -a $O/log.txt
fi
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline >> $O/bench_n1_runs.jsonl 2>>$O/bench_err.txt; done; echo "bench x3 rc=$?" | tee -a $O/log.txt
timeout 300 python tools/k2_time.py > $O/k2_time.txt 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/prof_run.txt 2>&1; echo "rocprof rc=$?" | tee -a $O/log.txt
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; cp $(find $O/prof -name "*domain_stats.csv" | head -1) $O/bench_domain_stats.csv 2>/dev/null
rm -rf $O/prof
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-prof > $O/pmc_run_$c.txt 2>&1; echo "pmc $c rc=$?" | tee -a $O/log.txt
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1); mv $f $O/pmc_$c/pmc_counter_collection.csv 2>/dev/null
done
python $R/tools/pmc_summary.py $O $O | tee -a $O/log.txt
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --output-format csv -d $O/pmc_sq -o pmc -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-prof > $O/pmc_run_sq.txt 2>&1; echo "pmc sq rc=$?" | tee -a $O/log.txt
python $R/tools/pmc_sq_summary.py $(find $O/pmc_sq -name "*counter_collection.csv" | head -1) > $O/pmc_sq_k1.txt 2>&1
rm -rf $O/pmc_sq
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k2prof -o k2 -- python $R/tools/k2_profile.py > $O/k2_run.txt 2>&1
python $R/tools/k2_profile.py --summarise $(find $O/k2prof -name "*kernel_stats.csv" | head -1) > $O/k2_kernels.txt 2>&1
rm -rf $O/k2prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mbprof -o mb -- python $R/bench.py --workload mobilenet_v1 --steps 20 --warmup 5 --no-cpu-baseline --no-prof > $O/mb_run.txt 2>&1
cp $(find $O/mbprof -name "*kernel_stats.csv" | head -1) $O/mobilenet_kernel_stats.csv; rm -rf $O/mbprof
# round 5: the row-streaming body alone (A/B against the bodies it replaces), the per-layer in-step table and the fp32 line
cd $R
timeout 400 python tools/rs_bench.py > $O/rs_bench.txt 2>&1; echo "rs_bench rc=$?" | tee -a $O/log.txt
# round 6: the channel-sliced single-pass backward alone (A/B), where an in-step kernel's time goes relative to alone
timeout 400 python tools/bs_bench.py > $O/bs_bench.txt 2>&1; echo "bs_bench rc=$?" | tee -a $O/log.txt
timeout 300 python tools/instep_probe.py > $O/instep_probe.txt 2>&1; echo "instep_probe rc=$?" | tee -a $O/log.txt
timeout 300 python tools/bnload_bench.py > $O/bnload_bench_kernels.txt 2>&1; echo "bnload_bench rc=$?" | tee -a $O/log.txt
timeout 300 python tools/bench_kernels.py --out $O/bk_warm.json > /dev/null 2>&1
timeout 300 python tools/bench_kernels.py --cold --out $O/bk_cold.json > /dev/null 2>&1
timeout 300 python tools/instep_table.py --alone $O/bk_warm.json --cold $O/bk_cold.json --out $O/instep_table.json > $O/instep_table.txt 2>&1; echo "instep rc=$?" | tee -a $O/log.txt
python tools/layer_excess.py $O/instep_table.json > $O/layer_excess.txt 2>&1
timeout 600 python bench.py --precision float32 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_fp32.json 2>>$O/bench_err.txt; echo "fp32 rc=$?" | tee -a $O/log.txt
cd $R; timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests_final.txt 2>&1; echo "pytest gpu rc=$?" | tee -a $O/log.txt; tail -3 $O/gpu_tests_final.txt | tee -a $O/log.txt
tail -3 $O/pmc_sq_k1.txt | tee -a $O/log.txt; tail -3 $O/k2_kernels.txt | tee -a $O/log.txt; tail -4 $O/bench_kernels_per_layer.txt | tee -a $O/log.txt
ls -la $O | tee -a $O/log.txt
