#!/bin/bash
# cfg5 (lock-step batch): host-side event sequence of the last batch steps
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
LBFGSX_HOST_TRACE=/tmp/ht5.txt python bench.py --workload cfg5-batched --steps 50 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
python - <<'P'
ev=[l.rstrip('\n').split(' ',1) for l in open('/tmp/ht5.txt')]
ev=[(int(t),g) for t,g in ev]
tail=ev[-48:]
pt=tail[0][0]
for t,g in tail:
    print('%8.1f  %s'%((t-pt)/1e3,g[:100])); pt=t
P
