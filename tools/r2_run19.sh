#!/bin/bash
mkdir -p gpurun_out/r2w
{
RIGL_BN_ORDER=7 timeout 600 python -m pytest tests/test_bn_gpu.py -q -m gpu 2>&1 | tail -2
export AB_STEPS=60 AB_WARMUP=15
bash tools/ab.sh "o0:RIGL_BN_ORDER=0" "o1:RIGL_BN_ORDER=1" "o2:RIGL_BN_ORDER=2" "o4:RIGL_BN_ORDER=4" "o3:RIGL_BN_ORDER=3" "o5:RIGL_BN_ORDER=5" "o6:RIGL_BN_ORDER=6" "o7:RIGL_BN_ORDER=7" "o0:RIGL_BN_ORDER=0"
} > gpurun_out/r2w/log.txt 2>&1
cat gpurun_out/r2w/log.txt
