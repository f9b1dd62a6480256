#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2o; mkdir -p $O
timeout 1200 python -m pytest tests/test_bn_gpu.py tests/test_e2e_gpu.py tests/test_k3_k1_gpu.py tests/test_k1_parity_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $O/log.txt
bash tools/ab.sh "off:RIGL_BN_FUSE_BWD=0" "mb8:RIGL_BN_FUSE_MAX_MB=8" "mb16:RIGL_BN_FUSE_MAX_MB=16" "mb30:RIGL_BN_FUSE_MAX_MB=30" "mb60:RIGL_BN_FUSE_MAX_MB=60" "all:RIGL_BN_FUSE_BWD=1" "off:RIGL_BN_FUSE_BWD=0" 2>&1 | tee -a $O/log.txt
