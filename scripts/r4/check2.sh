#!/bin/bash
# micro-benchmark of the sweep kernels, the L-BFGS-B tests, the per-m table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
scripts/experiments/kernels_x.bin > gpurun_out/r4/kernels_x_${TAG:-2}.txt 2>&1; cat gpurun_out/r4/kernels_x_${TAG:-2}.txt
timeout 1500 python -m pytest tests/test_lbfgsb_gpu.py tests/test_gcp_device_gpu.py -q --maxfail=10 > gpurun_out/r4/pytest_lbfgsb.log 2>&1
tail -15 gpurun_out/r4/pytest_lbfgsb.log
MS="${MS:-10 12 20}" bash scripts/r4/by_m.sh ${TAG:-split3}
