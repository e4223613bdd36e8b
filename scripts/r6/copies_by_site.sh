#!/bin/bash
# cfg4 from x0: which call sites issue the small copies of the first iterations (host trace, copies tagged copy@<line of lbfgsb.hip>)
cd $GRAFT_REPO_ROOT
LBFGSX_HOST_TRACE=/tmp/ht.txt python scripts/bench_lbfgsb.py --n 1e7 --iters ${1:-18} > /dev/null 2>&1
python - <<'PY'
import collections
ev=[l.rstrip("\n").split(" ",1) for l in open("/tmp/ht.txt")]
posts=[i for i,(t,g) in enumerate(ev) if "k_b_post" in g]
# the solve's iterations: the last len run (the warm-up solve comes first)
print("events", len(ev), "post launches", len(posts))
seg=ev[posts[-17]-1:] if len(posts)>=17 else ev
c=collections.Counter(g for t,g in seg if g.startswith("copy"))
for k,v in c.most_common(): print(v,k)
print("launches", sum(1 for t,g in seg if not g.startswith("copy") and g not in (">sync","<sync")), "syncs", sum(1 for t,g in seg if g==">sync"))
PY
