#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r2p; mkdir -p $O
timeout 600 python -m pytest tests/test_bn_gpu.py -x -q -m gpu -k "one_launch" 2>&1 | tail -12 | tee $O/log.txt
timeout 300 python tools/bnfuse_probe.py 2>&1 | tail -6 | tee -a $O/log.txt
RIGL_BN_COOP=0 timeout 300 python tools/bnfuse_probe.py 2>&1 | tail -6 | tee -a $O/log.txt
timeout 1200 python -m pytest tests/test_bn_gpu.py tests/test_e2e_gpu.py tests/test_configs_gpu.py tests/test_graphed_step_gpu.py tests/test_network_grad_gpu.py tests/test_fullsize_properties_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee -a $O/log.txt
bash tools/ab.sh "coop1:RIGL_BN_COOP=1" "coop0:RIGL_BN_COOP=0" "coop1:RIGL_BN_COOP=1" "coop0:RIGL_BN_COOP=0" 2>&1 | tee -a $O/log.txt
for w in mobilenet_v1 wrn22; do for c in 1 0; do
RIGL_BN_COOP=$c timeout 300 python bench.py --workload $w --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w coop=$c', round(d['value']), round(d['ms_per_step'],3))" | tee -a $O/log.txt
done; done
