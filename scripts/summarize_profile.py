#!/usr/bin/env python
"""Condense gpurun_out/prof_<round>/ (rocprofv3 kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes) into
profiles/<round>_*.  HBM bytes follow MI355X_MICROARCH.md "HBM": counters are in KiB and FETCH_SIZE
reports exactly half of a wide coalesced read stream on gfx950 (calibrated here on k_copy, whose byte
count is known), so  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections
import csv
import json
import os
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = os.path.join("gpurun_out", "prof_" + rnd)
os.makedirs("profiles", exist_ok=True)
shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join("profiles", rnd + "_kernel_stats.csv"))
for name in ("default", "cfg4_m15", "cfg4_m15_dd", "sharded_one_process_2x", "single_process_2x", "northstar", "northstar_unfused", "northstar_gram", "northstar_gram_f32h", "sharded_n1", "cfg3_m20", "cfg3_m20_gram", "cfg2_quad1e7", "cfg5_batched", "cfg4_lbfgsb",
             "cfg4_lbfgsb_mfma", "cfg4_lbfgsb_devmin4096", "cfg4_lbfgsb_i8", "cfg4_lbfgsb_scan", "two_ranks_one_device"):
    f = os.path.join(src, "bench_%s.json" % name)
    if os.path.exists(f):
        shutil.copy(f, os.path.join("profiles", "%s_bench_%s.json" % (rnd, name)))
for extra in ("cfg4_timeline.txt",):
    f = os.path.join(src, extra)
    if os.path.exists(f):
        shutil.copy(f, os.path.join("profiles", "%s_%s" % (rnd, extra)))
for sub in ("lbfgsb", "lbfgsb_mfma", "lbfgsb_i8", "lbfgsb_m15"):
    f = os.path.join(src, sub, "b_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join("profiles", "%s_%s_kernel_stats.csv" % (rnd, sub)))


def short(name):
    n = name.split("(")[0].replace("void lbfgsx::", "")
    return n


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for which in ("pmc_fetch", "pmc_write"):
    with open(os.path.join(src, which, "bench_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
stats = {}
with open(os.path.join(src, "trace", "bench_kernel_stats.csv")) as f:
    for r in csv.DictReader(f):
        stats[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]))

out = {"round": rnd, "n": 100000000, "m": 10,  # bench.py's defaults: the profiled command passes no --n / --m
       "command": "rocprofv3 {--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE} "
                                "--output-format csv -- python bench.py --no-cpu --steps 10 --warmup 11",
       "units": "FETCH_SIZE/WRITE_SIZE in KiB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction)",
       "kernels": {}}
for k, c in sorted(agg.items()):
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    fetch = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
    write = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
    hbm = (2.0 * fetch + write) * 1024.0
    calls, avg_ns = stats.get(k, (0, 0.0))
    out["kernels"][k] = {"calls": calls, "avg_ms": avg_ns * 1e-6, "fetch_KiB_raw": fetch, "write_KiB": write,
                         "hbm_bytes_per_launch": hbm,
                         "hbm_GBs": (hbm / (avg_ns * 1e-9) / 1e9) if avg_ns else None}
# launch-weighted two-loop figures (what bench.py's roofline object quotes)
tl = {k: v for k, v in out["kernels"].items() if k.startswith("k_twoloop") and "persist" not in k}
ps = {k: v for k, v in out["kernels"].items() if k.startswith("k_twoloop_persist")}
calls = sum(v["calls"] for v in tl.values())
def _fused_post(name):  # k_twoloop_persist<T, FUSE[, MEET]>: the second template argument
    args = name[name.index("<") + 1:name.rindex(">")].split(",") if "<" in name else []
    return len(args) >= 2 and args[1].strip() == "true"


fused = {k: v for k, v in ps.items() if _fused_post(k)}
if fused:
    # round 2: every apply_Hv after the first rides in the launch that also carries K3 (k_twoloop_persist<T, true>,
    # lbfgsx_post_linesearch_spec).  Launch k (k = 1, 2, ...) runs 2*min(k, 10)+1 steps, step 0 being the post statements.
    v = list(fused.values())[0]
    steps = sum(2 * min(k, 10) + 1 for k in range(1, v["calls"] + 1))
    out["fused_post"] = True
    out["twoloop_persistent"] = {"kernel": list(fused.keys())[0], "launches": v["calls"], "steps": steps,
                                 "hbm_bytes_per_step": v["hbm_bytes_per_launch"] * v["calls"] / steps,
                                 "ms_per_step": v["avg_ms"] * v["calls"] / steps,
                                 "algorithmic_bytes_per_step_full_history": (8 * 10 + 5) * 1e8 * 8 / 21.0}
    out["twoloop_avg_hbm_bytes_per_launch"] = out["twoloop_persistent"]["hbm_bytes_per_step"]
    out["twoloop_avg_ms"] = out["twoloop_persistent"]["ms_per_step"]
elif ps:
    # one persistent launch per apply_Hv = 2*ncorr+1 steps; the profiled command starts from an empty history with
    # m = 10, so launch k (k = 0, 1, ...) runs 2*min(k, 10)+1 steps.  Per-step figures make it comparable with the
    # step-wise kernels and with bench.py's algorithmic bytes per step.
    v = list(ps.values())[0]
    steps = sum(2 * min(k, 10) + 1 for k in range(v["calls"]))
    out["twoloop_persistent"] = {"launches": v["calls"], "steps": steps,
                                 "hbm_bytes_per_step": v["hbm_bytes_per_launch"] * v["calls"] / steps,
                                 "ms_per_step": v["avg_ms"] * v["calls"] / steps}
    out["twoloop_avg_hbm_bytes_per_launch"] = out["twoloop_persistent"]["hbm_bytes_per_step"]
    out["twoloop_avg_ms"] = out["twoloop_persistent"]["ms_per_step"]
elif calls:
    out["twoloop_avg_hbm_bytes_per_launch"] = sum(v["hbm_bytes_per_launch"] * v["calls"] for v in tl.values()) / calls
    out["twoloop_avg_ms"] = sum(v["avg_ms"] * v["calls"] for v in tl.values()) / calls
with open(os.path.join("profiles", rnd + "_pmc_summary.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))
for k, v in out["kernels"].items():
    print("%-40s calls %4d avg %.4f ms  hbm %.4g B  %.0f GB/s" % (k[:40], v["calls"], v["avg_ms"], v["hbm_bytes_per_launch"], v["hbm_GBs"] or 0))


# ---- the opt-in Gram-space recursion (f64 history and f32 history): same counters for its two kernels
def gram_summary(prefix, tag, flag, kprefix="k_gs_", note=None):
    gsrc = os.path.join(src, prefix + "_trace", "bench_kernel_stats.csv")
    if not os.path.exists(gsrc):
        return
    shutil.copy(gsrc, os.path.join("profiles", "%s_%s_kernel_stats.csv" % (rnd, tag)))
    gagg = collections.defaultdict(lambda: collections.defaultdict(list))
    for which in (prefix + "_pmc_fetch", prefix + "_pmc_write"):
        fn = os.path.join(src, which, "bench_counter_collection.csv")
        if os.path.exists(fn):
            with open(fn) as f:
                for r in csv.DictReader(f):
                    gagg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    gstats = {}
    with open(gsrc) as f:
        for r in csv.DictReader(f):
            gstats[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]))
    gout = {"round": rnd, "n": 100000000, "m": 10,  # bench.py's defaults: the profiled command passes no --n / --m
       "command": out["command"].replace("bench.py", "bench.py " + flag), "units": out["units"],
            "note": note or "launches start from an empty history (m = 10): launch k reads 2*min(k,10) columns, so the per-launch "
                    "averages below mix the warm-up launches with the full-history ones; max_* are the full-history launches",
            "kernels": {}}
    for k, c in sorted(gagg.items()):
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c or not k.startswith(kprefix):
            continue
        hb = [(2.0 * fv + wv) * 1024.0 for fv, wv in zip(c["FETCH_SIZE"], c["WRITE_SIZE"])]
        calls, avg_ns = gstats.get(k, (0, 0.0))
        gout["kernels"][k] = {"calls": calls, "avg_ms": avg_ns * 1e-6, "hbm_bytes_per_launch_avg": sum(hb) / len(hb),
                              "hbm_bytes_per_launch_max": max(hb)}
    with open(os.path.join("profiles", "%s_%s_pmc_summary.json" % (rnd, tag)), "w") as f:
        json.dump(gout, f, indent=1)
    for k, v in gout["kernels"].items():
        print("%-40s calls %4d avg %.4f ms  hbm avg %.4g B max %.4g B" % (k[:40], v["calls"], v["avg_ms"],
                                                                          v["hbm_bytes_per_launch_avg"], v["hbm_bytes_per_launch_max"]))


gram_summary("gram", "gram", "--recursion gram")
gram_summary("f32h", "gram_f32h", "--recursion gram-f32h")
# cfg5: 1024 lock-step f32 problems of n = 1e5 (bench.py --workload cfg5-batched --steps 50)
gram_summary("batched", "batched", "--workload cfg5-batched --steps 50 --no-cpu", kprefix="kb_",
             note="one launch covers the 1024 problems of the batch; launches start from an empty history (m = 10), "
                  "max_* are the full-history launches: kb_twoloop_full reads 4m columns-worth of f32 per problem "
                  "(the direction vector stays on the CU), i.e. 1024 * (40 + 2) * 1e5 * 4 B = 17.2 GB algorithmic")


# ---- cfg4 (L-BFGS-B): per-kernel HBM traffic of scripts/bench_lbfgsb.py --n 1e7 --iters 40
def lbfgsb_pmc():
    tr = os.path.join(src, "lbfgsb", "b_kernel_stats.csv")
    ff = os.path.join(src, "lbfgsb_pmc_fetch", "b_counter_collection.csv")
    fw = os.path.join(src, "lbfgsb_pmc_write", "b_counter_collection.csv")
    if not (os.path.exists(tr) and os.path.exists(ff) and os.path.exists(fw)):
        return
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in (ff, fw):
        with open(fn) as f:
            for r in csv.DictReader(f):
                a[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    st = {}
    with open(tr) as f:
        for r in csv.DictReader(f):
            st[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"]))
    tot = sum(v[2] for v in st.values())
    o = {"round": rnd, "command": "rocprofv3 {--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE} --output-format csv -- "
                                  "python scripts/bench_lbfgsb.py --n 1e7 --iters 40  (cfg4; includes its untimed warm-up solve at n = 2^18)",
         "units": out["units"], "total_kernel_ms": tot * 1e-6, "kernels": {}}
    for k, c in a.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c or k not in st:
            continue
        calls, avg_ns, total_ns = st[k]
        hb = (2.0 * sum(c["FETCH_SIZE"]) + sum(c["WRITE_SIZE"])) * 1024.0   # all launches of the kernel
        o["kernels"][k] = {"calls": calls, "avg_ms": avg_ns * 1e-6, "share_of_kernel_time": total_ns / tot,
                           "hbm_bytes_total": hb, "hbm_GBs": hb / (total_ns * 1e-9) / 1e9 if total_ns else None}
    o["kernels"] = dict(sorted(o["kernels"].items(), key=lambda kv: -kv[1]["share_of_kernel_time"]))
    with open(os.path.join("profiles", rnd + "_lbfgsb_pmc_summary.json"), "w") as f:
        json.dump(o, f, indent=1)
    for k, v in list(o["kernels"].items())[:14]:
        print("%-44s calls %4d avg %.4f ms  %5.1f %%  %.0f GB/s" % (k[:44], v["calls"], v["avg_ms"], 100 * v["share_of_kernel_time"], v["hbm_GBs"] or 0))


lbfgsb_pmc()
