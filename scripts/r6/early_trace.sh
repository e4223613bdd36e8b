#!/bin/bash
# cfg4 from x0: iterations A..B of the solve in order (launches, waits; copies summarised), LBFGSX_HOST_TRACE
cd $GRAFT_REPO_ROOT
LBFGSX_HOST_TRACE=/tmp/ht.txt python scripts/bench_lbfgsb.py --n 1e7 --m ${1:-10} --iters 20 --no-warmup > /dev/null 2>&1
A=${2:-3} B=${3:-6} python - <<'PY'
import os
ev=[l.rstrip("\n").split(" ",1) for l in open("/tmp/ht.txt")]
ev=[(int(t),g) for t,g in ev]
posts=[i for i,(t,g) in enumerate(ev) if "k_b_post" in g]
A,B=int(os.environ["A"]),int(os.environ["B"])
a,b=posts[A-1],posts[B]
t0=ev[a][0]; prev=t0; ncopy=0
for t,g in ev[a:b+1]:
    if g==">sync": prev=t; continue
    if g.startswith("copy@"): ncopy+=1; continue
    if g.startswith("lbfgsb:"): continue
    print("%8.1f us  (+%6.1f)  %s%s" % ((t-t0)/1e3,(t-prev)/1e3,g[:100], ("   [%d copies]"%ncopy) if ncopy else ""))
    ncopy=0; prev=t
PY
