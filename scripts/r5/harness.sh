#!/bin/bash
# the passes of cfg4 on their own (scripts/experiments/kernels_x.hip): product build, then the builds with parts of
# kx_solve_sweep switched off (LBFGSX_X_DBG: 16 no sweep statements, 32 no y / rhs stores, 64 no double-double products,
# 128 no left-to-right sums, 256 no list appends, 496 all of them)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
O=gpurun_out/r5/kernels_x_${TAG:-a}.txt
( echo "== product build"; timeout 120 scripts/experiments/kernels_x.bin
  for d in 16 32 64 128 256 496; do echo "== LBFGSX_X_DBG=$d"; KX_QUICK=1 timeout 60 scripts/experiments/kernels_x_dbg$d.bin | grep "grid  512"; done ) > $O 2>&1
cat $O
