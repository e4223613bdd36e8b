#!/bin/bash
# ROCTx ranges of the solvers' phases beside the kernels: LBFGSX_ROCTX=1 under rocprofv3 --marker-trace --kernel-trace (no counters)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r6/markers; rm -rf $O; mkdir -p $O
LBFGSX_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $O/cfg4 -o b -- python scripts/bench_lbfgsb.py --n 1e7 --iters 40 > $O/cfg4.log 2>&1
LBFGSX_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $O/cfg5 -o b -- python bench.py --workload cfg5-batched --steps 50 --no-cpu > $O/cfg5.log 2>&1
for t in cfg4 cfg5; do echo "== $t"; f=$(find $O/$t -name "*marker_api_stats.csv" | head -1); [ -n "$f" ] && head -12 $f | cut -c1-160; find $O/$t -name "*trace.csv" -delete; done
