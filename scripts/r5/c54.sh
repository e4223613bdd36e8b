#!/bin/bash
# the (5, 4) class against (10, 2) at 2c = 20 (scripts/experiments/kernels_x.hip, KX_54=1)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
O=gpurun_out/r5/kernels_x_c54_${TAG:-a}.txt
( for r in 1 2; do KX_54=1 KX_QUICK=1 timeout 120 scripts/experiments/kernels_x.bin | grep "^split\|^----"; done ) > $O 2>&1
cat $O
