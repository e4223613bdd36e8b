#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
KX_QUICK=1 scripts/experiments/kernels_x.bin > gpurun_out/r4/kernels_x_${TAG:-3}.txt 2>&1; cat gpurun_out/r4/kernels_x_${TAG:-3}.txt
MS="${MS:-10 20}" bash scripts/r4/by_m.sh ${TAG:-3}
timeout 2400 python -m pytest tests -m gpu -q --maxfail=15 > gpurun_out/r4/pytest_all.log 2>&1
tail -25 gpurun_out/r4/pytest_all.log
