#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2f; mkdir -p $O
L="g3c2=14,14,256,256,3,1 g4c2=7,7,512,512,3,1 g2c2=28,28,128,128,3,1"
echo "c3    : $(python tools/layer_probe.py --what fwd $L 2>/dev/null)" | tee $O/abl.txt
for v in 1 2 4 8 6; do
  echo "abl$v  : $(RIGL_HIP_LIB=$PWD/build/alt/librigl_c3abl$v.so python tools/layer_probe.py --what fwd $L 2>/dev/null)" | tee -a $O/abl.txt
done
