#!/bin/bash
set -u
R="$(cd "$(dirname "$0")/.." && pwd)"
O=$R/gpurun_out/r2e; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
L="g3c2=14,14,256,256,3,1 g3c1=14,14,1024,256,1,1"
for m in 2 0; do
  RIGL_T196=$m RIGL_W9=0 timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d $O/pmc_t196_$m -o pmc -- python $R/tools/layer_probe.py --what fwd $L > $O/pmc_run_$m.txt 2>&1
  f=$(find $O/pmc_t196_$m -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_kernel_summary.py $f > $O/pmc_summary_t196_$m.txt 2>&1
  rm -rf $O/pmc_t196_$m
done
cat $O/pmc_summary_t196_2.txt $O/pmc_summary_t196_0.txt
