#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2m; mkdir -p $O
timeout 900 python -m pytest tests/test_k2_prune_regrow_gpu.py tests/test_tf2_golden.py tests/test_mask_updaters.py tests/test_tf_random.py -x -q -m gpu 2>&1 | tail -8 | tee $O/log.txt
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_configs_gpu.py tests/test_fullsize_properties_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee -a $O/log.txt
echo "== v2" | tee -a $O/log.txt; timeout 300 python tools/k2_time.py 2>&1 | tail -4 | tee -a $O/log.txt
echo "== v1" | tee -a $O/log.txt; RIGL_K2_V1=1 timeout 300 python tools/k2_time.py 2>&1 | tail -4 | tee -a $O/log.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o k2 -- python $R/tools/k2_profile.py > $R/$O/prof_run.txt 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/k2_kernel_stats.csv; find $O/prof -name "*.csv" ! -name "*stats*" -delete
python tools/k2_profile.py --summarise $O/k2_kernel_stats.csv | tee -a $O/log.txt
