#!/bin/bash
# the north-star's timed iterations, in order: launches and waits with the host time between them (LBFGSX_HOST_TRACE)
cd $GRAFT_REPO_ROOT
LBFGSX_HOST_TRACE=/tmp/htn.txt python bench.py --no-cpu --no-batched --no-legs --steps 6 --warmup 11 $NS_ARGS > /dev/null 2>&1
python - <<'PY'
ev=[l.rstrip("\n").split(" ",1) for l in open("/tmp/htn.txt")]
ev=[(int(t),g) for t,g in ev]
per=[i for i,(t,g) in enumerate(ev) if "k_twoloop_persist" in g]
a,b=per[-4],per[-2]
t0=ev[a][0]; prev=t0
for t,g in ev[a:b+1]:
    print("%9.1f us  (+%8.1f)  %s" % ((t-t0)/1e3,(t-prev)/1e3,g[:100])); prev=t
PY
