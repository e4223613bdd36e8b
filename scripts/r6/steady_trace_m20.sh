#!/bin/bash
# cfg4 at m = 20: the last four steady iterations in order (launches, copies, waits; LBFGSX_HOST_TRACE)
cd $GRAFT_REPO_ROOT
LBFGSX_HOST_TRACE=/tmp/ht.txt python scripts/bench_lbfgsb.py --n 1e7 --m ${1:-20} --iters ${2:-60} > /dev/null 2>&1
python - <<'PY'
ev=[l.rstrip("\n").split(" ",1) for l in open("/tmp/ht.txt")]
ev=[(int(t),g) for t,g in ev]
posts=[i for i,(t,g) in enumerate(ev) if "k_b_post" in g]
a,b=posts[-6],posts[-2]
t0=ev[a][0]
prev=t0
for t,g in ev[a:b+1]:
    if g in (">sync",): 
        prev=t; continue
    print("%8.1f us  (+%6.1f)  %s" % ((t-t0)/1e3,(t-prev)/1e3,g[:110]))
    prev=t
PY
