#!/bin/bash
# Round-2 baseline on one box: GPU test suite (with durations), bench line, rocprof kernel stats.
set -u
R="$(cd "$(dirname "$0")/.." && pwd)"
O=$R/gpurun_out/r2base; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee $O/log.txt; tail -22 $O/pytest_gpu.txt | tee -a $O/log.txt
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/bench.json 2>$O/bench_err.txt; echo "bench rc=$?" | tee -a $O/log.txt; tail -1 $O/bench.json | cut -c1-1500 | tee -a $O/log.txt
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_run.txt 2>&1; echo "rocprof rc=$?" | tee -a $O/log.txt
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/bench_kernel_stats.csv 2>/dev/null
find $O/prof -name "*.csv" ! -name "*stats*" -delete 2>/dev/null
head -40 $O/bench_kernel_stats.csv | cut -c1-200
