#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2aa
{
cd $R
timeout 300 python -m pytest tests/test_bn_gpu.py -q -m gpu -k "two_batch or bottleneck" 2>&1 | tail -15
timeout 300 python -m pytest tests/test_e2e_gpu.py tests/test_network_grad_gpu.py tests/test_graphed_step_gpu.py -q -m gpu -x 2>&1 | tail -3
export AB_STEPS=80 AB_WARMUP=15
bash tools/ab.sh "off:RIGL_BN_PAIR=0" "on:RIGL_BN_PAIR=1" "off:RIGL_BN_PAIR=0" "on:RIGL_BN_PAIR=1" "off:RIGL_BN_PAIR=0" "on:RIGL_BN_PAIR=1"
} > $R/gpurun_out/r2aa/log.txt 2>&1
cat $R/gpurun_out/r2aa/log.txt
