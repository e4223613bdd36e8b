#!/usr/bin/env python
"""The host side of the benchmark's first L-BFGS-B iteration (LBFGSX_HOST_TRACE file): every interval above 100 us between two
consecutive host events of the big run before its second post launch, and the totals per kind.
    LBFGSX_HOST_TRACE=/tmp/ht.txt python scripts/bench_lbfgsb.py --n 1e7 --iters 3; python scripts/r5/first_iter_host.py /tmp/ht.txt"""
import collections
import sys
ev = []
for line in open(sys.argv[1]):
    t, tag = line.rstrip("\n").split(" ", 1)
    ev.append((int(t), tag))
posts = [i for i, (_, g) in enumerate(ev) if "k_b_post" in g]
nbig = int(sys.argv[2]) if len(sys.argv) > 2 else 3
first_big = posts[-nbig]
# the big run starts at the generator launch before it
start = max(i for i in range(first_big) if "gen" in ev[i][1])
seg = ev[start:first_big + 1]
print("events %d, wall %.2f ms" % (len(seg), (seg[-1][0] - seg[0][0]) / 1e6))
tot = collections.Counter()
in_sync = 0
prev = seg[0]
for t, g in seg[1:]:
    dt = t - prev[0]
    if g == "<sync":
        in_sync += dt
    else:
        tot[g.split("<")[0][:40]] += dt
    if dt > 100e3:
        print("%9.1f us at %7.2f ms: %s  ->  %s" % (dt / 1e3, (t - seg[0][0]) / 1e6, prev[1][:60], g[:60]))
    prev = (t, g)
print("in waits %.2f ms" % (in_sync / 1e6))
for k, v in tot.most_common(12):
    print("%8.2f ms host time before  %s" % (v / 1e6, k))
