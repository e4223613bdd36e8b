#!/bin/bash
# kernel trace of cfg4 (m from $M) and its per-iteration timeline (scripts/trace_cfg4.py) -> gpurun_out/r5/cfg4_timeline_m$M.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
M=${M:-10}; IT=40; [ $M -gt 16 ] && IT=60
O=gpurun_out/r5/tl_m$M; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python scripts/bench_lbfgsb.py --n 1e7 --m $M --iters $IT > $O.log 2>&1
TL_ITERS=$IT python scripts/trace_cfg4.py $O > gpurun_out/r5/cfg4_timeline_m$M.txt 2>&1
find $O -name "*.csv" -delete
head -60 gpurun_out/r5/cfg4_timeline_m$M.txt | cut -c1-220
