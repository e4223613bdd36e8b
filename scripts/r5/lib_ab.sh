#!/bin/bash
# A/B of two builds of liblbfgsx.so on one box, interleaved (the box's copy of the repo is scratch: the library file is swapped):
# every wait of the steady cfg4 iteration + the from-x0 rate; $1 = the other library
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
cp lbfgspp_amd/liblbfgsx.so /tmp/lib_head.so; cp $1 /tmp/lib_other.so
for rep in 1 2 3; do
for which in head other; do
  cp /tmp/lib_$which.so lbfgspp_amd/liblbfgsx.so
  LBFGSX_HOST_TRACE=/tmp/ht.txt python scripts/bench_lbfgsb.py --n 1e7 --m ${M:-10} --iters 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$which  from x0 %.1f it/s  steady(median) %.1f' % (d['it_per_s'], d['steady_it_per_s']))"
  python scripts/host_trace.py /tmp/ht.txt | grep -A9 "^--- waits" | grep -v "^---" | awk -v c="$which" '{printf "%-6s %8.1f us/it %5.2f/it %8.1f avg  %s\n", c, $1, $2, $3, $4}'
done
done | tee gpurun_out/r5/lib_ab_${TAG:-a}.txt
cp /tmp/lib_head.so lbfgspp_amd/liblbfgsx.so
