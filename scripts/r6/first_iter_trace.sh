#!/bin/bash
# cfg4 from x0: the first iteration of the timed solve (everything before its first post-line-search launch), waits and gaps above 50 us
cd $GRAFT_REPO_ROOT
LBFGSX_HOST_TRACE=/tmp/ht.txt python scripts/bench_lbfgsb.py --n 1e7 --m ${1:-10} --iters 3 > /dev/null 2>&1
python - <<'PY'
ev=[l.rstrip("\n").split(" ",1) for l in open("/tmp/ht.txt")]
ev=[(int(t),g) for t,g in ev]
posts=[i for i,(t,g) in enumerate(ev) if "k_b_post" in g]
# the timed solve is the last one: its first post launch is posts[-2] (3 iterations -> 2 posts per solve... take the last two)
b=posts[-2]
# walk back to the solve's first evaluation
a=b
while a>0 and "k_b_eval" not in ev[a][1]: a-=1
t0=ev[a][0]; prev=t0; n=0; ncopy=0
for t,g in ev[a:b+1]:
    d=(t-prev)/1e3
    if g.startswith("copy@"): ncopy+=1
    if d>50 or g.startswith("lbfgsb:") or "k_b_" in g or "k_cauchy" in g:
        print("%9.1f us  (+%8.1f)  %s%s" % ((t-t0)/1e3,d,g[:90], ("  [%d copies since]"%ncopy) if ncopy else "")); ncopy=0
    prev=t
print("first iteration: %.2f ms, %d events" % ((ev[b][0]-t0)/1e6, b-a))
PY
