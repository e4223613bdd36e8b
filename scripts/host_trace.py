#!/usr/bin/env python
"""Where the host spends its time between two device waits (LBFGSX_HOST_TRACE=<file>; ctx.hpp).

    LBFGSX_HOST_TRACE=/tmp/ht.txt python scripts/bench_lbfgsb.py --n 1e7 --iters 40
    python scripts/host_trace.py /tmp/ht.txt [first_iteration last_iteration]

Iterations are delimited by the k_b_post launches.  For every launch that follows a wait: the host time from the end of the
wait to the launch (what the device sits idle for, besides the launch latency); for every wait: how long it lasted.
"""
import collections
import sys

ev = []
for line in open(sys.argv[1]):
    t, tag = line.rstrip("\n").split(" ", 1)
    ev.append((int(t), tag))
posts = [i for i, (_, g) in enumerate(ev) if "k_b_post" in g]
posts = posts[-40:]
lo_it = int(sys.argv[2]) if len(sys.argv) > 2 else len(posts) - 13
hi_it = int(sys.argv[3]) if len(sys.argv) > 3 else len(posts) - 1
seg = ev[posts[lo_it]:posts[hi_it]]
nit = hi_it - lo_it
wall = (ev[posts[hi_it]][0] - ev[posts[lo_it]][0]) / nit
host = collections.OrderedDict()
waits = collections.OrderedDict()
last_wait_end = None
last_launch = None
wait_start = None
in_wait = 0
for t, g in seg:
    if g == ">sync":
        wait_start = t
        continue
    if g == "<sync":
        a = waits.setdefault(last_launch or "?", [0, 0])
        a[0] += 1
        a[1] += t - wait_start
        in_wait += t - wait_start
        last_wait_end = t
        continue
    if last_wait_end is not None:
        a = host.setdefault(g, [0, 0])
        a[0] += 1
        a[1] += t - last_wait_end
        last_wait_end = None
    if not g.startswith("copy"):
        last_launch = g
print("iterations %d..%d: wall %.3f ms/it, in waits %.3f ms/it, host outside waits %.3f ms/it"
      % (lo_it + 1, hi_it, wall / 1e6, in_wait / nit / 1e6, (wall - in_wait / nit) / 1e6))
print("\n--- host time from the end of a wait to the next launch/copy, by what is launched (us/it, count/it, avg us) ---")
for g, (c, tt) in sorted(host.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%8.1f  %5.2f  %8.1f  %s" % (tt / nit / 1e3, c / nit, tt / c / 1e3, g[:90]))
# every interval between two consecutive events outside the waits, charged to the later event
gapsum = collections.OrderedDict()
prev_t = None
for t, g in seg:
    if g == "<sync":
        prev_t = t
        continue
    if prev_t is not None:
        a = gapsum.setdefault(g, [0, 0])
        a[0] += 1
        a[1] += t - prev_t
    prev_t = t
print("\n--- host time before each event (from the previous event or the end of a wait), by event (us/it, count/it, avg us) ---")
for g, (c, tt) in sorted(gapsum.items(), key=lambda kv: -kv[1][1])[:24]:
    print("%8.1f  %5.2f  %8.1f  %s" % (tt / nit / 1e3, c / nit, tt / c / 1e3, g[:90]))
print("\n--- waits, by the last kernel launched before them (us/it, count/it, avg us) ---")
for g, (c, tt) in sorted(waits.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%8.1f  %5.2f  %8.1f  %s" % (tt / nit / 1e3, c / nit, tt / c / 1e3, g[:90]))

# one iteration, event by event (the last one of the range): time since the previous event
print("\n--- the last iteration of the range, event by event (us since the previous event) ---")
one = ev[posts[hi_it - 1]:posts[hi_it]]
pt = one[0][0]
for t, g in one:
    print("%8.1f  %s" % ((t - pt) / 1e3, g[:100]))
    pt = t
