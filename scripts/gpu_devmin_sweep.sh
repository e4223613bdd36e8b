#!/bin/bash
# cfg4 from x0 against the hand-over point of the Cauchy search (host form below it, device form + host chain above)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 0 32 64 128 256 512; do
LBFGSX_GCP_DEVICE_MIN=$v python scripts/bench_lbfgsb.py --n 1e7 --iters 40 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('devmin $v: from x0 %.1f steady %.1f fx %.17g gcp_total_us %d' % (d['it_per_s'], d['steady_it_per_s'], d.get('fx', 0), d['stats']['gcp_total_us']))"
done; done
