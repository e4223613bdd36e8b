#!/bin/bash
# the launches between the post pass and the W'd pass (candidate sort, first chunk of the search) under the settings given as
# arguments (VAR=value strings): the wait that follows kx_multidot2_wf covers them and the pass itself (scripts/host_trace.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
for rep in 1 2 3; do
for cfg in "$@"; do
  env $cfg LBFGSX_HOST_TRACE=/tmp/ht.txt python scripts/bench_lbfgsb.py --n 1e7 --m ${M:-10} --iters 40 > /dev/null 2>&1
  python scripts/host_trace.py /tmp/ht.txt | grep -A14 "^--- waits" | grep "multidot2_wf" | sed "s/^/$cfg  /"
done
done | tee gpurun_out/r5/chain_ab_${TAG:-a}.txt
