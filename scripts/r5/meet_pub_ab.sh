#!/bin/bash
# the persistent launch's meeting points under two settings A="$1" B="$2" (VAR=value strings; default LBFGSX_MEET_PUB=1 / 0: the
# polled word published before / after the dot's copy for the host), interleaved on one box: cfg2 (n = 1e7, where a meeting point is a fifth of a step), the north-star
# and cfg3; the L-BFGS tests first (bit-identical results either way: the knob moves no arithmetic)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
if [ -z "$NOTESTS" ]; then
  timeout 1200 python -m pytest tests/test_lbfgs_gpu.py tests/test_full_size_gpu.py -m gpu -q --maxfail=5 2>&1 | tail -3
fi
A=${1:-LBFGSX_MEET_PUB=1}; B=${2:-LBFGSX_MEET_PUB=0}
OUT=gpurun_out/r5/meet_pub_ab_${TAG:-a}.txt; : > $OUT
run () {
  name=$1; pub=$2; shift 2
  env $pub python bench.py --no-cpu --no-batched --no-legs --verbose "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-24s %-10s %8.1f it/s  ms/step %.3f  frac %.3f  apply_Hv %.3f ms  avg launch-step %.4f ms' % ('$pub', '$name', d['value'], d['ms_per_step'], r['frac'], r['apply_Hv_ms'], r['avg_launch_ms']))" >> $OUT
}
run cfg2 "$A" --objective quadratic --n 10000000 > /dev/null; : > $OUT   # warm the box
for rep in $(seq 1 ${ROUNDS:-4}); do
  for pub in "$A" "$B"; do run cfg2 "$pub" --objective quadratic --n 10000000; done
done
for rep in 1 2; do
  for pub in "$A" "$B"; do run north-star "$pub"; done
done
cat $OUT
