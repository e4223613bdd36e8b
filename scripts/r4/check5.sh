#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_lbfgsb_gpu.py tests/test_gcp_device_gpu.py -q --maxfail=10 > gpurun_out/r4/pytest_lbfgsb.log 2>&1
tail -12 gpurun_out/r4/pytest_lbfgsb.log
TAG=${TAG:-5} bash scripts/r4/ab_env.sh ${ABENV:-LBFGSX_RHS_IDENTITY=0}
TAG=${TAG:-5} bash scripts/r4/timeline.sh | head -30
