#!/bin/bash
# same-box A/B of cfg4 knobs + the L-BFGS-B GPU tests + a kernel timeline of the default configuration
# usage: gpu_ab_cfg4.sh KNOB [pytest -k expression]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
K=${1:-LBFGSX_WTD_COMPACT}
O=gpurun_out/ab_cfg4; rm -rf $O; mkdir -p $O
run() { env "$@" python scripts/bench_lbfgsb.py --n 1e7 --iters 40 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*: from x0 %.1f steady %.1f sweeps %d fx %.17g' % (d['it_per_s'], d['steady_it_per_s'], d['stats']['submin_sweeps'], d.get('fx', 0)))"; }
for rep in 1 2; do
run $K=0
run $K=1
done
timeout 900 python -m pytest tests/test_lbfgsb_gpu.py tests/test_gcp_device_gpu.py -x -q ${2:+-k "$2"} > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
rocprofv3 --kernel-trace --output-format csv -d $O/t -o b -- python scripts/bench_lbfgsb.py --n 1e7 --iters 40 > $O/run.json 2> $O/run.err
python scripts/trace_cfg4.py $O/t > $O/summary.txt 2>&1
find $O -name "*.csv" -delete
head -32 $O/summary.txt
