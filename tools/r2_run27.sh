#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2cc
{
cd $R
export AB_STEPS=80 AB_WARMUP=15
bash tools/ab.sh "w2:RIGL_BN_APPLY_WIDE=2" "w4:RIGL_BN_APPLY_WIDE=4" "w8:RIGL_BN_APPLY_WIDE=8" "w2:RIGL_BN_APPLY_WIDE=2" "w4:RIGL_BN_APPLY_WIDE=4" "w8:RIGL_BN_APPLY_WIDE=8"
} > $R/gpurun_out/r2cc/log.txt 2>&1
cat $R/gpurun_out/r2cc/log.txt
