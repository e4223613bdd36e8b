#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
REPS=3 bash scripts/r5/quick.sh
rm -rf /tmp/tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o b -- python scripts/bench_lbfgsb.py --n 1e7 --m 10 --iters 40 > /dev/null 2>&1
python scripts/trace_cfg4.py /tmp/tl 12 > gpurun_out/r5/cfg4_timeline_m10_2.txt 2>&1
head -3 gpurun_out/r5/cfg4_timeline_m10_2.txt; sed -n '/before and including/,$p' gpurun_out/r5/cfg4_timeline_m10_2.txt
timeout 900 python -m pytest tests/test_lbfgsb_gpu.py tests/test_gcp_device_gpu.py tests/test_edge_cases_gpu.py tests/test_lbfgs_gpu.py -m gpu -q -x 2>&1 | tail -4
