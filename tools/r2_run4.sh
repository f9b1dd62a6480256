#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r2d; mkdir -p $O
RIGL_T196=2 RIGL_W9=0 timeout 900 python tests/k1_check.py --set t196 > $O/k1_t196.txt 2>&1; echo "k1_check t196 rc=$?" | tee $O/log.txt; tail -3 $O/k1_t196.txt | tee -a $O/log.txt
L="g3c2=14,14,256,256,3,1 g4c2=7,7,512,512,3,1 g2c2=28,28,128,128,3,1 g3c1=14,14,1024,256,1,1 g4c1=7,7,2048,512,1,1"
echo "old   : $(RIGL_T196=0 RIGL_C3=0 RIGL_W9=0 python tools/layer_probe.py --what fwd,dgrad $L 2>/dev/null)" | tee -a $O/log.txt
echo "new   : $(RIGL_T196=2 RIGL_C3=1 RIGL_W9=0 python tools/layer_probe.py --what fwd,dgrad $L 2>/dev/null)" | tee -a $O/log.txt
