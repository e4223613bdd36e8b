#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/subset; mkdir -p $O
timeout 1200 python -m pytest "$@" -x -q --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -32 $O/pytest.log
