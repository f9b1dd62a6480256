#!/bin/bash
# HBM traffic of the depthwise kernels in the MobileNet-v1 step (two PMC passes)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2hh; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python $R/bench.py --workload mobilenet_v1 --steps 4 --warmup 2 --no-cpu-baseline --no-prof > $O/run_$c.txt 2>&1; echo "pmc $c rc=$?"
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1); mv $f $O/pmc_$c/pmc_counter_collection.csv 2>/dev/null
done
python - <<PY
import csv, collections
def load(path, col):
    d=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(path)):
        n=r['Kernel_Name'].split('(')[0].replace('void ','')
        d[n][0]+=1; d[n][1]+=float(r['Counter_Value'])
    return d
f=load('$O/pmc_FETCH_SIZE/pmc_counter_collection.csv','FETCH_SIZE'); w=load('$O/pmc_WRITE_SIZE/pmc_counter_collection.csv','WRITE_SIZE')
rows=[]
for n in f:
    if 'kdw' in n or 'kbn' in n:
        rows.append((n, f[n][0], f[n][1]*1024*2/f[n][0]/1e6, w.get(n,[1,0])[1]*1024/max(w.get(n,[1,0])[0],1)/1e6))
for r in sorted(rows, key=lambda r:-r[1]*(r[2]+r[3]))[:16]:
    print('%-60s launches %4d  read %8.1f MB  write %8.1f MB per launch'%r)
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
