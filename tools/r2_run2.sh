#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r2b; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/probes/dma_rate_probe.hip -o /tmp/dma_probe 2>/dev/null
timeout 300 /tmp/dma_probe > $O/dma_probe.txt 2>&1; echo "probe rc=$?" | tee $O/log.txt
echo "== k1_check t196 set with wgrad9 (RIGL_T196=0 RIGL_W9=1)" | tee -a $O/log.txt
RIGL_T196=0 RIGL_W9=1 timeout 900 python tests/k1_check.py --set t196 > $O/k1_w9.txt 2>&1; echo "rc=$?" | tee -a $O/log.txt; tail -4 $O/k1_w9.txt | tee -a $O/log.txt
echo "== bench_kernels RIGL_T196=0 RIGL_W9=1" | tee -a $O/log.txt
RIGL_T196=0 RIGL_W9=1 timeout 600 python tools/bench_kernels.py --out $O/bench_kernels_w9.json > $O/bench_kernels_w9.txt 2>&1; echo "rc=$?" | tee -a $O/log.txt
grep -E "c2 |TOTAL" $O/bench_kernels_w9.txt | awk '!s[$2$3$4$5$6$7]++' | tee -a $O/log.txt
