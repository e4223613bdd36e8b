#!/bin/bash
# host-side timeline of cfg4 (where the host spends the time between two device waits), steady and early iterations
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/host_trace; rm -rf $O; mkdir -p $O
LBFGSX_HOST_TRACE=/tmp/ht.txt python scripts/bench_lbfgsb.py --n 1e7 --iters 40 2>/dev/null | tail -1 > $O/run.json
python scripts/host_trace.py /tmp/ht.txt > $O/steady.txt
python scripts/host_trace.py /tmp/ht.txt 0 17 > $O/early.txt
head -3 $O/steady.txt; grep -A75 "the last iteration of the range" $O/steady.txt
