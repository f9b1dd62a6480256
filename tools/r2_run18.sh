#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2r; mkdir -p $O
timeout 900 python -m pytest tests/test_bn_gpu.py tests/test_e2e_gpu.py tests/test_configs_gpu.py tests/test_graphed_step_gpu.py tests/test_network_grad_gpu.py tests/test_fullsize_properties_gpu.py -x -q -m gpu 2>&1 | tail -6 | tee $O/log.txt
bash tools/ab.sh "lazy1:RIGL_BN_LAZY_DRES=1" "lazy0:RIGL_BN_LAZY_DRES=0" "lazy1:RIGL_BN_LAZY_DRES=1" "lazy0:RIGL_BN_LAZY_DRES=0" 2>&1 | tee -a $O/log.txt
