#!/bin/bash
mkdir -p gpurun_out/r2q
{
export AB_ARGS="--workload mobilenet_v1" AB_STEPS=100 AB_WARMUP=20
bash tools/ab.sh "off:RIGL_DW_STATS=0" "on1:RIGL_DW_STATS_REPS=1" "on2:RIGL_DW_STATS_REPS=2" "on3:RIGL_DW_STATS_REPS=3" "off:RIGL_DW_STATS=0" "on1:RIGL_DW_STATS_REPS=1" "on2:RIGL_DW_STATS_REPS=2" "on3:RIGL_DW_STATS_REPS=3"
} > gpurun_out/r2q/log.txt 2>&1
cat gpurun_out/r2q/log.txt
