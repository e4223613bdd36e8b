#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
for d in 0 1 2 4 8 3 15; do echo "== LBFGSX_X_DBG=$d (1: no chains, 2: no dd products, 4: no stores, 8: no shuffles)"; KX_QUICK=1 scripts/experiments/kernels_x_dbg$d.bin | grep split; done > gpurun_out/r4/kernels_x_dbg.txt 2>&1
cat gpurun_out/r4/kernels_x_dbg.txt
