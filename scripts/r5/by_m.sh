#!/bin/bash
# cfg4's problem (n = 1e7) at several history lengths: steady / from-x0 rates and the byte model's fraction (scripts/r5/quick.sh per m)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
OUT=gpurun_out/r5/by_m_${TAG:-a}.txt
: > $OUT
for m in ${MS:-10 12 16 20 40}; do M=$m REPS=2 bash scripts/r5/quick.sh >> $OUT 2>&1; done
cat $OUT
