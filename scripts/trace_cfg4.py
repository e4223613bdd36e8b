#!/usr/bin/env python
"""Timeline of a steady cfg4 (L-BFGS-B) iteration from a rocprofv3 kernel trace: per kernel and per gap.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o b -- python scripts/bench_lbfgsb.py --n 1e7 --iters 40
    python scripts/trace_cfg4.py DIR > summary.txt

Iterations are delimited by k_b_post launches (one per iteration); the last `tail` iterations are averaged.
"""
import collections
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(n):
    n = n.replace("void ", "").replace("lbfgsx::", "")
    n = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", n)
    n = n.split("(")[0]
    return n[:70]


posts = [i for i, r in enumerate(rows) if "k_b_post" in r[2]]
# the warm-up solve comes first: keep the posts of the big run = the last 40
posts = posts[-(int(os.environ.get("TL_ITERS", "40"))):]
lo, hi = posts[-tail - 1], posts[-1]
seg = rows[lo:hi]
wall = (rows[hi][0] - rows[lo][0]) / tail
busy = sum(e - s for s, e, _ in seg) / tail
agg = collections.OrderedDict()
for s, e, n in seg:
    k = short(n)
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += e - s
print("steady iteration (last %d): wall %.3f ms, kernels %.3f ms, idle %.3f ms, launches %.1f"
      % (tail, wall / 1e6, busy / 1e6, (wall - busy) / 1e6, len(seg) / tail))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%7.3f ms/it  %5.2f calls/it  %8.1f us avg  %s" % (t / tail / 1e6, c / tail, t / c / 1e3, k))
# one iteration in order, with the idle gap before each kernel
print("\n--- the last iteration, in order (gap before the kernel, duration) ---")
s0 = posts[-2]
prev_end = rows[s0][0]
for s, e, n in rows[s0:posts[-1]]:
    print("%8.1f us gap %8.1f us  %s" % ((s - prev_end) / 1e3, (e - s) / 1e3, short(n)))
    prev_end = e
# the first iterations: where the from-x0 time goes
print("\n--- per-iteration wall (ms), all 40 ---")
print(" ".join("%.2f" % ((rows[posts[i + 1]][0] - rows[posts[i]][0]) / 1e6) for i in range(len(posts) - 1)))

# the first iterations one by one: wall, kernel time, launches, the three largest kernels
print("\n--- the first 20 iterations: wall ms | kernels ms | launches | largest kernels (ms) ---")
for i in range(min(20, len(posts) - 1)):
    seg_i = rows[posts[i]:posts[i + 1]]
    w = (rows[posts[i + 1]][0] - rows[posts[i]][0]) / 1e6
    busy_i = sum(e - s for s, e, _ in seg_i) / 1e6
    ag = collections.Counter()
    for s_, e_, n_ in seg_i:
        ag[short(n_)[:34]] += (e_ - s_) / 1e6
    top = ", ".join("%s %.2f" % (k, v) for k, v in ag.most_common(4))
    print("%2d  %6.2f | %6.2f | %4d | %s" % (i + 1, w, busy_i, len(seg_i), top))

# the early iterations together: kernel totals, and where the device waits for the host
early = min(17, len(posts) - 1)
seg_e = rows[posts[0]:posts[early]]
wall_e = (rows[posts[early]][0] - rows[posts[0]][0]) / 1e6
busy_e = sum(e - s for s, e, _ in seg_e) / 1e6
print("\n--- iterations 1..%d together: wall %.2f ms, kernels %.2f ms, idle %.2f ms, launches %d ---"
      % (early, wall_e, busy_e, wall_e - busy_e, len(seg_e)))
ag = collections.OrderedDict()
for s, e, n in seg_e:
    a = ag.setdefault(short(n), [0, 0])
    a[0] += 1
    a[1] += e - s
for k, (c, t) in sorted(ag.items(), key=lambda kv: -kv[1][1])[:28]:
    print("%8.3f ms  %5d calls  %8.1f us avg  %s" % (t / 1e6, c, t / c / 1e3, k))
gaps = collections.OrderedDict()
prev = None
for s, e, n in seg_e:
    if prev is not None:
        g = s - prev[1]
        if g > 8000:
            a = gaps.setdefault((short(prev[2])[:40], short(n)[:40]), [0, 0])
            a[0] += 1
            a[1] += g
    prev = (s, e, n)
print("\n--- gaps above 8 us in iterations 1..%d, by (kernel before -> kernel after): total ms, count, avg us ---" % early)
for (a_, b_), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%8.3f ms  %5d  %8.1f us  %s -> %s" % (t / 1e6, c, t / c / 1e3, a_, b_))

# single early iterations, kernel by kernel
for it in (1, 2, 5, 10):
    if it >= len(posts):
        continue
    seg_i = rows[posts[it - 1]:posts[it]]
    ag = collections.OrderedDict()
    for s, e, n in seg_i:
        a = ag.setdefault(short(n), [0, 0])
        a[0] += 1
        a[1] += e - s
    gap_i = 0
    big = []
    prev = None
    for s, e, n in seg_i:
        if prev is not None and s - prev[1] > 8000:
            gap_i += s - prev[1]
            big.append(((s - prev[1]) / 1e3, short(prev[2])[:30], short(n)[:30]))
        prev = (s, e, n)
    print("\n--- iteration %d: %d launches, gaps above 8 us %.2f ms ---" % (it, len(seg_i), gap_i / 1e6))
    for k, (c, t) in sorted(ag.items(), key=lambda kv: -kv[1][1])[:16]:
        print("%8.3f ms  %5d calls  %8.1f us avg  %s" % (t / 1e6, c, t / c / 1e3, k))
    big.sort(reverse=True)
    print("   largest gaps: " + "; ".join("%.0f us %s -> %s" % g for g in big[:8]))

# Round 5: what the benchmark calls the FIRST iteration -- everything of the big run before its second post-line-search
# launch (the initial evaluation, the first Cauchy search over all n break points, the first line search with its 20 trials,
# the first post) -- which the iteration boundaries above leave out.  The big run starts at the first kernel after the
# warm-up solve's last post.
allposts = [i for i, r in enumerate(rows) if "k_b_post" in r[2]]
first_big = allposts[-(int(os.environ.get("TL_ITERS", "40")))]
prev_small = [i for i in allposts if i < first_big]
start = (prev_small[-1] + 1) if prev_small else 0
# skip the tail of the warm-up solve (its last line search etc.): the big run begins with the generator / fill kernels
for i in range(start, first_big):
    if "k_gen" in rows[i][2] or "fill" in rows[i][2].lower():
        start = i
        break
seg0 = rows[start:first_big + 1]
if seg0:
    wall0 = (rows[first_big][1] - seg0[0][0]) / 1e6
    busy0 = sum(e - s for s, e, _ in seg0) / 1e6
    print("\n--- before and including the first post launch of the big run: wall %.2f ms, kernels %.2f ms, idle %.2f ms, %d launches ---"
          % (wall0, busy0, wall0 - busy0, len(seg0)))
    agg0 = collections.OrderedDict()
    for s, e, n in seg0:
        a = agg0.setdefault(short(n), [0, 0])
        a[0] += 1
        a[1] += e - s
    for k, (c, t) in sorted(agg0.items(), key=lambda kv: -kv[1][1])[:16]:
        print("%8.3f ms  %5d calls  %8.1f us avg  %s" % (t / 1e6, c, t / c / 1e3, k))
    gaps0 = []
    prev_end = seg0[0][0]
    for s, e, n in seg0:
        if s - prev_end > 20e3:
            gaps0.append(((s - prev_end) / 1e3, (s - seg0[0][0]) / 1e6, short(n)))
        prev_end = max(prev_end, e)
    print("gaps above 20 us (us, at ms, before kernel):")
    for g in sorted(gaps0, reverse=True)[:14]:
        print("  %9.1f us at %7.2f ms before %s" % g)
