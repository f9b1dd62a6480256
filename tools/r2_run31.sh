#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2gg
{
cd $R
export AB_STEPS=80 AB_WARMUP=15
bash tools/ab.sh "old:RIGL_HIP_LIB=$R/build/alt/librigl_convold.so" "new:" "old:RIGL_HIP_LIB=$R/build/alt/librigl_convold.so" "new:" "old:RIGL_HIP_LIB=$R/build/alt/librigl_convold.so" "new:"
} > $R/gpurun_out/r2gg/log.txt 2>&1
cat $R/gpurun_out/r2gg/log.txt
