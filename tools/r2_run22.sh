#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2z
{
cd $R
timeout 200 python -m pytest tests/test_bn_gpu.py -q -m gpu -k "maxpool" 2>&1 | tail -5
export AB_STEPS=80 AB_WARMUP=15
bash tools/ab.sh "off:RIGL_STEM_TAIL=0" "on:RIGL_STEM_TAIL=1" "off:RIGL_STEM_TAIL=0" "on:RIGL_STEM_TAIL=1"
} > $R/gpurun_out/r2z/log.txt 2>&1
cat $R/gpurun_out/r2z/log.txt
rm -f $R/gpurun_out/r2z/kern.txt
sed -i 's/for v in 1 0; do/for v in 1; do/' $R/tools/r2_run23.sh
bash $R/tools/r2_run23.sh
