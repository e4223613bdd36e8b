#!/bin/bash
# host time between a wait and the launch that follows it, steady iterations of cfg4 (LBFGSX_HOST_TRACE + scripts/host_trace.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
M=${M:-10}
LBFGSX_HOST_TRACE=/tmp/ht.txt python scripts/bench_lbfgsb.py --n 1e7 --m $M --iters 40 > /dev/null 2>&1
python scripts/host_trace.py /tmp/ht.txt > gpurun_out/r5/host_gaps_m$M.txt 2>&1
cat gpurun_out/r5/host_gaps_m$M.txt
