#!/bin/bash
# round 5, first GPU call: the L-BFGS-B tests (changed kernels), the bench line's contract, the pass harness, the cfg4 timeline
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests/test_lbfgsb_gpu.py tests/test_gcp_device_gpu.py -m gpu -q -x --durations=8 -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r5/pytest_lbfgsb_1.log
tail -12 gpurun_out/r5/pytest_lbfgsb_1.log
timeout 400 python -m pytest tests/test_bench_contract_gpu.py -m gpu -q -x -k "default_line" 2>&1 | tail -15 > gpurun_out/r5/pytest_bench_1.log
tail -8 gpurun_out/r5/pytest_bench_1.log
TAG=1 bash scripts/r5/harness.sh > /dev/null 2>&1
head -40 gpurun_out/r5/kernels_x_1.txt
rm -rf /tmp/tl; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o b -- python scripts/bench_lbfgsb.py --n 1e7 --m 10 --iters 40 > gpurun_out/r5/bench_lbfgsb_prof_1.json 2>/dev/null
python scripts/trace_cfg4.py /tmp/tl 12 > gpurun_out/r5/cfg4_timeline_m10_1.txt 2>&1
head -60 gpurun_out/r5/cfg4_timeline_m10_1.txt
python scripts/bench_lbfgsb.py --n 1e7 --m 10 --iters 40 > gpurun_out/r5/bench_lbfgsb_1.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r5/bench_lbfgsb_1.json')); print({k:d.get(k) for k in ('it_per_s','steady_it_per_s','fx')}, d.get('model'), d['per_iter_ms'][:4])"
