#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2x
{
cd $R
export AB_STEPS=80 AB_WARMUP=15
bash tools/ab.sh "f16:RIGL_HIP_LIB=$R/build/alt/librigl_f16.so" "f32:" "f16:RIGL_HIP_LIB=$R/build/alt/librigl_f16.so" "f32:" "f16:RIGL_HIP_LIB=$R/build/alt/librigl_f16.so" "f32:"
} > $R/gpurun_out/r2x/log.txt 2>&1
cat $R/gpurun_out/r2x/log.txt
