#!/bin/bash
# kx_solve_sweep with two register sets against three (scripts/experiments/kernels_x.hip built with -DLBFGSX_X_NBUF=2 / 3)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
O=gpurun_out/r5/kernels_x_nbuf_${TAG:-a}.txt
( for r in 1 2; do for nb in 2 3; do echo "== NBUF=$nb"; timeout 120 scripts/experiments/kernels_x_nb$nb.bin | grep "^split"; done; done ) > $O 2>&1
cat $O
