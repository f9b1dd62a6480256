#!/bin/bash
# Round-end check on one box: the whole GPU suite, smoke(), then the profile set (gpurun_out/r2prof)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2final
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r2final/gpu_tests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 >> gpurun_out/r2final/gpu_tests.txt
cat gpurun_out/r2final/gpu_tests.txt
timeout 1500 bash tools/r2_profiles.sh 2>&1 | tail -30
