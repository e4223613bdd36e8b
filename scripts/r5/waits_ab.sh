#!/bin/bash
# every wait of the steady cfg4 iteration (us per iteration, by the last kernel launched before it: scripts/host_trace.py) under
# the settings given as arguments (VAR=value strings), interleaved
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
for rep in $(seq 1 ${ROUNDS:-2}); do
for cfg in "$@"; do
  env $cfg LBFGSX_HOST_TRACE=/tmp/ht.txt python scripts/bench_lbfgsb.py --n 1e7 --m ${M:-10} --iters 40 > /dev/null 2>&1
  python scripts/host_trace.py /tmp/ht.txt | grep -A9 "^--- waits" | grep -v "^---" | awk -v c="$cfg" '{printf "%-22s %8.1f us/it %5.2f/it %8.1f avg  %s\n", c, $1, $2, $3, $4}'
  echo
done
done | tee gpurun_out/r5/waits_ab_${TAG:-a}.txt
