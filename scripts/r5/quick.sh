#!/bin/bash
# quick same-box check of a kernel change: cfg4 (m from $M, default 10) REPS times -- steady / from-x0 rates, fx, the byte
# model's fraction -- then the tests named in $TESTS (pytest -k expression over the L-BFGS-B files), if any
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
M=${M:-10}; IT=40; [ $M -gt 16 ] && IT=60
for r in $(seq 1 ${REPS:-3}); do
  python scripts/bench_lbfgsb.py --n 1e7 --m $M --iters $IT 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); m=d.get('model') or {}
print('m=%d steady(median) %.1f it/s  window mean %.3f ms (%.1f it/s)  from x0 %.1f it/s  first %.1f ms  fx %.12g  model frac %.3f  syncs %.2f launches %.1f' % (d['m'], d['steady_it_per_s'], m.get('window_ms_per_iteration',0), 1e3/max(m.get('window_ms_per_iteration',1e9),1e-9), d['it_per_s'], d['per_iter_ms'][0], d['fx'], m.get('frac',0), m.get('host_syncs_per_iteration',0), m.get('launches_per_iteration',0)))"
done
if [ -n "$TESTS" ]; then
  timeout 900 python -m pytest tests/test_lbfgsb_gpu.py tests/test_gcp_device_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x -k "$TESTS" 2>&1 | tail -4
fi
