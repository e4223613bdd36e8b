#!/bin/bash
# cfg2 (L-BFGS, diag quadratic n = 1e7, m = 10, Nocedal-Wright): host-side event sequence of the last iterations
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
LBFGSX_HOST_TRACE=/tmp/ht2.txt python bench.py --n 1e7 --objective quadratic --no-cpu --no-legs --no-batched 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
python - <<'P'
ev=[l.rstrip('\n').split(' ',1) for l in open('/tmp/ht2.txt')]
ev=[(int(t),g) for t,g in ev]
tail=ev[-60:]
pt=tail[0][0]
for t,g in tail:
    print('%8.1f  %s'%((t-pt)/1e3,g[:90])); pt=t
P
