#!/bin/bash
# per-iteration timeline of cfg4 (m from $M, default 10) under the kernel trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
M=${M:-10}; IT=40; [ $M -gt 16 ] && IT=60
rm -rf /tmp/tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o b -- python scripts/bench_lbfgsb.py --n 1e7 --m $M --iters $IT > /tmp/tl.json 2>/dev/null
python scripts/trace_cfg4.py /tmp/tl 12 > gpurun_out/r4/cfg4_timeline_m${M}_${TAG:-a}.txt 2>&1
head -75 gpurun_out/r4/cfg4_timeline_m${M}_${TAG:-a}.txt
