#!/bin/bash
# HIP API time of a short cfg4 run (allocations and frees inside minimize)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/hipapi; rm -rf $O; mkdir -p $O
rocprofv3 --hip-trace --stats --output-format csv -d $O -o b -- python scripts/bench_lbfgsb.py --n 1e7 --m ${M:-10} --iters ${IT:-4} --no-warmup > $O.log 2>&1
f=$(find $O -name "*hip_api_stats.csv" | head -1)
head -25 $f | cut -c1-160
t=$(find $O -name "*hip_api_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
big=[r for r in rows if r['Function'] in ('hipMalloc','hipHostMalloc','hipFree','hipHostFree','hipMemset','hipMemsetAsync','hipDeviceSynchronize')]
t0=min(int(r['Start_Timestamp']) for r in rows)
for r in big:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    if d>100: print('%-16s at %9.2f ms  %9.1f us' % (r['Function'], (int(r['Start_Timestamp'])-t0)/1e6, d))
PY
find $O -name "*.csv" -size +1M -delete
