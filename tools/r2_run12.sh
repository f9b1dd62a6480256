#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2l; mkdir -p $O
timeout 900 python -m pytest tests/test_network_grad_gpu.py -x -q -m gpu -s 2>&1 | grep -v Warning | tail -60 | tee $O/log.txt
timeout 900 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee -a $O/log.txt
