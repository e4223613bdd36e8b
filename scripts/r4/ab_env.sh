#!/bin/bash
# A/B of environment knobs on one box: scripts/r4/ab_env.sh "LBFGSX_TRIAL_AHEAD=0" "LBFGSX_POST_BUILD=0" ... (each against the default)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
OUT=gpurun_out/r4/ab_env_${TAG:-1}.txt; : > $OUT
run () {  # run <label> <env assignment or empty>
  env $2 python scripts/bench_lbfgsb.py --n 1e7 --iters ${ITERS:-40} --m ${M:-10} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-28s from x0 %7.1f it/s  steady %7.1f it/s  fx %r' % ('$1', d['it_per_s'], d['steady_it_per_s'], d['fx']))" >> $OUT
}
for rep in 1 2 3; do
  run default ""
  for kv in "$@"; do run "$kv" "$kv"; done
done
cat $OUT
