#!/usr/bin/env python
"""Round 4: condense the rocprofv3 PMC passes of the legs (scripts/profile_r4.sh -> gpurun_out/prof_r4/) into
  profiles/r4_legs_pmc_summary.json   cfg4 (m = 10 and m = 20): HBM bytes per L-BFGS-B iteration, steady window and from x0;
                                      cfg5: HBM bytes per problem-iteration  -- what bench.py's leg_traffic() reads
  profiles/r4_cfg2_pmc_summary.json,  r4_cfg3_pmc_summary.json: HBM bytes per two-loop step (bench.py's pmc_traffic())
HBM bytes follow MI355X_MICROARCH.md "HBM": counters in KiB, FETCH_SIZE reports half of a wide coalesced read stream on
gfx950, so hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024; FETCH_SIZE and WRITE_SIZE come from SEPARATE passes of the same
(deterministic) command, matched dispatch by dispatch."""
import collections
import csv
import json
import os
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else os.path.join("gpurun_out", "prof_r4")
RND = sys.argv[2] if len(sys.argv) > 2 else "r4"
UNITS = "FETCH_SIZE/WRITE_SIZE in KiB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction)"


def short(name):
    return name.split("(")[0].replace("void lbfgsx::", "").replace("lbfgsx::", "")


def rows(path):
    """[(dispatch id, kernel, value)] of one single-counter pass, in dispatch order"""
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), float(r["Counter_Value"])))
    # one row per (dispatch, counter [, dimension]): add the dimensions up
    agg = collections.OrderedDict()
    for d, k, v in sorted(out, key=lambda t: t[0]):
        if d not in agg:
            agg[d] = [k, 0.0]
        agg[d][1] += v
    return [(d, kv[0], kv[1]) for d, kv in agg.items()]


def hbm_by_dispatch(prefix, base):
    f = rows(os.path.join(SRC, prefix + "_pmc_fetch", base + "_counter_collection.csv"))
    w = rows(os.path.join(SRC, prefix + "_pmc_write", base + "_counter_collection.csv"))
    if len(f) != len(w) or any(a[1] != b[1] for a, b in zip(f, w)):
        raise SystemExit("%s: the FETCH_SIZE and WRITE_SIZE passes did not dispatch the same kernels (%d vs %d)" % (prefix, len(f), len(w)))
    return [(a[1], (2.0 * a[2] + b[2]) * 1024.0) for a, b in zip(f, w)]


def stats(prefix, base):
    st = {}
    p = os.path.join(SRC, prefix, base + "_kernel_stats.csv")
    if os.path.exists(p):
        with open(p) as f:
            for r in csv.DictReader(f):
                st[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"]))
    return st


def cfg4(prefix, n, m, iters, warm_iters=12):
    """scripts/bench_lbfgsb.py --n N --m M --iters I: an untimed warm-up solve (12 iterations at n = 2^18) first, then the run.
    k_b_post runs exactly once per iteration (after the line search), so its dispatches delimit the iterations."""
    d = hbm_by_dispatch(prefix, "b")
    ends = [i for i, (k, _) in enumerate(d) if k.startswith("k_b_post")]
    if len(ends) != warm_iters + iters:
        raise SystemExit("%s: %d k_b_post dispatches, expected %d" % (prefix, len(ends), warm_iters + iters))
    run0 = ends[warm_iters - 1] + 1                 # first dispatch after the warm-up solve's last iteration
    w0 = iters // 2
    win0 = ends[warm_iters + w0 - 1] + 1            # bench.py's window: iterations w0+1 .. iters
    last = ends[-1] + 1
    tot_run = sum(v for _, v in d[run0:last])
    tot_win = sum(v for _, v in d[win0:last])
    per_kernel = collections.defaultdict(lambda: [0, 0.0])
    for k, v in d[win0:last]:
        per_kernel[k][0] += 1
        per_kernel[k][1] += v
    st = stats(prefix, "b")
    top = sorted(per_kernel.items(), key=lambda kv: -kv[1][1])[:14]
    return {"leg": "cfg4", "n": n, "m": m, "hbm_bytes": tot_win / (iters - w0), "per": "L-BFGS-B iteration, mean over the steady window "
            "(iterations %d..%d of %d from x0), every dispatch between two k_b_post launches counted" % (w0 + 1, iters, iters),
            "hbm_bytes_from_x0": tot_run / iters,
            "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of scripts/bench_lbfgsb.py --n %g --m %d --iters %d, taken separately"
                   % (n, m, iters),
            "window_dispatches_per_iteration": (last - win0) / float(iters - w0),
            "window_top_kernels": {k: {"launches_per_iteration": c / float(iters - w0), "hbm_bytes_per_launch": b / c,
                                       "avg_ms": (st[k][1] * 1e-6 if k in st else None)} for k, (c, b) in top}}


def cfg5(prefix, n, m, problems, iters):
    d = hbm_by_dispatch(prefix, "bench")
    tot = sum(v for k, v in d if k.startswith("kb_"))
    # round 6: the leg's command minimises the same batch several times (warm-ups, the timed solve, the instrumented one);
    # every minimisation evaluates the start points once (kb_eval)
    solves = max(1, sum(1 for k, _ in d if k.startswith("kb_eval")))
    full = [v for k, v in d if k.startswith("kb_iter") or k.startswith("kb_twoloop_full")]
    return {"leg": "cfg5", "n": n, "m": m, "hbm_bytes": tot / float(problems * iters * solves),
            "per": "problem-iteration, mean over the %d lock-step iterations from x0 of %d problems (history filling during the first %d), "
                   "%d identical minimisations in the profiled command; all kb_* launches counted" % (iters, problems, m, solves),
            "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py --workload cfg5-batched --steps %d --no-cpu, taken separately" % iters,
            "one_launch_iteration_full_history_hbm_bytes_per_problem": (max(full) / problems) if full else None}


def twoloop(prefix, n, m, tag, command):
    """the fused persistent launch: launch k (k = 1, 2, ...) runs 2 min(k, m) + 1 steps, step 0 being the post statements"""
    d = hbm_by_dispatch(prefix, "bench")
    fused = [(k, v) for k, v in d if k.startswith("k_twoloop_persist") and "true" in k.split(",")[1]]
    if not fused:
        raise SystemExit(prefix + ": no fused persistent launches")
    steps = sum(2 * min(k, m) + 1 for k in range(1, len(fused) + 1))
    st = stats(prefix, "bench")
    name = fused[0][0]
    out = {"round": RND, "n": n, "m": m, "fused_post": True, "command": command, "units": UNITS,
           "twoloop_persistent": {"kernel": name, "launches": len(fused), "steps": steps,
                                  "hbm_bytes_per_step": sum(v for _, v in fused) / steps,
                                  "hbm_bytes_full_history_launch": max(v for _, v in fused),
                                  "ms_per_step": (st[name][2] * 1e-6 / steps) if name in st else None,
                                  "algorithmic_bytes_per_step_full_history": (8 * m + 5) * n * 8 / (2.0 * m + 1)}}
    out["twoloop_avg_hbm_bytes_per_launch"] = out["twoloop_persistent"]["hbm_bytes_per_step"]
    out["twoloop_avg_ms"] = out["twoloop_persistent"]["ms_per_step"]
    with open(os.path.join("profiles", "%s_%s_pmc_summary.json" % (RND, tag)), "w") as f:
        json.dump(out, f, indent=1)
    print(tag, json.dumps(out["twoloop_persistent"]))


legs = []
for prefix, n, m, iters in (("cfg4_m10", 10_000_000, 10, 40), ("cfg4_m20", 10_000_000, 20, 60)):
    if os.path.exists(os.path.join(SRC, prefix + "_pmc_fetch")):
        legs.append(cfg4(prefix, n, m, iters))
if os.path.exists(os.path.join(SRC, "batched_pmc_fetch")):
    legs.append(cfg5("batched", 100_000, 10, 1024, 50))
with open(os.path.join("profiles", RND + "_legs_pmc_summary.json"), "w") as f:
    json.dump({"round": RND, "units": UNITS, "legs": legs}, f, indent=1)
for e in legs:
    print(e["leg"], e["n"], e["m"], "%.4g B per %s" % (e["hbm_bytes"], e["per"][:40]), e.get("hbm_bytes_from_x0"))
if os.path.exists(os.path.join(SRC, "cfg2_pmc_fetch")):
    twoloop("cfg2", 10_000_000, int(os.environ.get("CFG2_M", "10")), "cfg2",
            "rocprofv3 {--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE} -- python bench.py --no-cpu --no-batched --no-legs --objective quadratic --n 10000000")
if os.path.exists(os.path.join(SRC, "cfg3_pmc_fetch")):
    twoloop("cfg3", 100_000_000, 20, "cfg3",
            "rocprofv3 {--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE} -- python bench.py --no-cpu --no-batched --no-legs --m 20 --steps 10 --warmup 22")
