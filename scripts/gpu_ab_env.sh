#!/bin/bash
# same-box A/B of one environment knob on cfg4 (n = 1e7, m = 10, 40 iterations): KNOB=0 and KNOB=1 alternating, 3 rounds
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
K=$1
run() { env "$@" python scripts/bench_lbfgsb.py --n 1e7 --iters 40 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*: from x0 %.1f steady %.1f sweeps %d fx %.17g polled %s timeouts %s' % (d['it_per_s'], d['steady_it_per_s'], d['stats']['submin_sweeps'], d.get('fx', 0), d.get('polled_waits'), d.get('poll_timeouts')))"; }
for rep in 1 2 3; do
run $K=0
run $K=1
done
