#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2y
{
cd $R
export AB_STEPS=80 AB_WARMUP=15
AB_ARGS="" bash tools/ab.sh "eager:"
AB_ARGS="--graph" bash tools/ab.sh "graph:"
AB_ARGS="" bash tools/ab.sh "eager:"
AB_ARGS="--graph" bash tools/ab.sh "graph:"
} > $R/gpurun_out/r2y/log.txt 2>&1
cat $R/gpurun_out/r2y/log.txt
