#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2z
cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
RIGL_STEM_TAIL=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$v -o st -- python $R/bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-prof > /dev/null 2>&1
f=$(find /tmp/prof$v -name "*kernel_stats.csv" | head -1)
python - "$f" $v <<'PY' >> $R/gpurun_out/r2z/kern.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print('RIGL_STEM_TAIL='+sys.argv[2])
for r in rows:
    n=r['Name']
    if 'kpool' in n or ('kbn' in n and int(r['Calls'])<=40) :
        print('  %-90s calls %4s avg %7.1f us'%(n[:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
cat $R/gpurun_out/r2z/kern.txt
