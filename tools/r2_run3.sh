#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r2c; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/probes/dma_rate_probe.hip -o /tmp/dma_probe 2>/dev/null
timeout 300 /tmp/dma_probe > $O/dma_probe.txt 2>&1
L="g3c1=14,14,1024,256,1,1 g3c3=14,14,256,1024,1,1 g3c2=14,14,256,256,3,1 g4c2=7,7,512,512,3,1 g2c2=28,28,128,128,3,1 g4c1=7,7,2048,512,1,1"
echo "old   : $(RIGL_T196=0 RIGL_W9=0 python tools/layer_probe.py --what fwd $L 2>/dev/null)" | tee $O/abl.txt
echo "t196  : $(RIGL_T196=2 RIGL_W9=0 python tools/layer_probe.py --what fwd $L 2>/dev/null)" | tee -a $O/abl.txt
for v in 1 2 4 8 3; do
  echo "abl$v  : $(RIGL_HIP_LIB=$PWD/build/alt/librigl_abl$v.so RIGL_T196=2 RIGL_W9=0 python tools/layer_probe.py --what fwd $L 2>/dev/null)" | tee -a $O/abl.txt
done
