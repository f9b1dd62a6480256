#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2n; mkdir -p $O
timeout 900 python -m pytest tests/test_bn_gpu.py -x -q -m gpu 2>&1 | tail -12 | tee $O/log.txt
timeout 1200 python -m pytest tests/test_e2e_gpu.py tests/test_configs_gpu.py tests/test_k3_k1_gpu.py tests/test_graphed_step_gpu.py tests/test_network_grad_gpu.py -x -q -m gpu 2>&1 | tail -8 | tee -a $O/log.txt
bash tools/ab.sh "fuse1:RIGL_BN_FUSE_BWD=1" "fuse0:RIGL_BN_FUSE_BWD=0" "fuse1:RIGL_BN_FUSE_BWD=1" "fuse0:RIGL_BN_FUSE_BWD=0" 2>&1 | tee -a $O/log.txt
