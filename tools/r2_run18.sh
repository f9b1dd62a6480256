#!/bin/bash
mkdir -p gpurun_out/r2v
timeout 900 python tools/vendor_compare.py --find 1 --out gpurun_out/r2v/vendor_find.json > gpurun_out/r2v/find.txt 2>&1
echo "find rc=$?"; tail -60 gpurun_out/r2v/find.txt
