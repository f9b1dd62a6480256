#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2bb
{
cd $R
timeout 300 python -m pytest tests/test_k3_k1_gpu.py -q -m gpu -k "depthwise" 2>&1 | tail -12
timeout 300 python -m pytest tests/test_configs_gpu.py -q -m gpu -k "mobilenet or Mobile" 2>&1 | tail -4
export AB_ARGS="--workload mobilenet_v1" AB_STEPS=100 AB_WARMUP=20
bash tools/ab.sh "off:RIGL_DW_BN=0" "on:RIGL_DW_BN=1" "off:RIGL_DW_BN=0" "on:RIGL_DW_BN=1"
} > $R/gpurun_out/r2bb/log.txt 2>&1
cat $R/gpurun_out/r2bb/log.txt
