#!/bin/bash
# round 4, first check of the split-row kernels: the L-BFGS-B tests, then the per-m table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_lbfgsb_gpu.py tests/test_gcp_device_gpu.py -q --maxfail=25 -k "${K:-}" > gpurun_out/r4/pytest_lbfgsb.log 2>&1
tail -40 gpurun_out/r4/pytest_lbfgsb.log
MS="${MS:-10 12 16 20}" bash scripts/r4/by_m.sh ${TAG:-split1}
