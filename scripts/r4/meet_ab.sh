#!/bin/bash
# the two forms of the persistent launch's meeting points, same box: tests first, then cfg2 / north-star / cfg3 A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_lbfgs_gpu.py tests/test_full_size_gpu.py tests/test_edge_cases_gpu.py -q --maxfail=10 > gpurun_out/r4/pytest_lbfgs.log 2>&1
tail -12 gpurun_out/r4/pytest_lbfgs.log
OUT=gpurun_out/r4/meet_ab_${TAG:-1}.txt; : > $OUT
for rep in 1 2; do
for meet in all last; do
  for cfg in "cfg2 --objective quadratic --n 10000000" "north-star" "cfg3 --m 20 --steps 10 --warmup 22"; do
    set -- $cfg; name=$1; shift
    LBFGSX_MEET=$meet python bench.py --no-cpu --no-batched --no-legs "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('LBFGSX_MEET=%-4s %-10s %8.1f it/s  ms/step %.3f  frac %.3f  apply_Hv %.3f ms  avg launch-step %.4f ms' % ('$meet', '$name', d['value'], d['ms_per_step'], r['frac'], r['apply_Hv_ms'], r['avg_launch_ms']))" >> $OUT
  done
done
done
cat $OUT
