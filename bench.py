#!/usr/bin/env python
"""bench.py -- L-BFGS iterations/s on MI355X (BASELINE.json metric) + roofline + CPU baseline.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full L-BFGS iteration (line search with the fused trial kernel, the post-line-search
pass, history commit and the 2m+1-launch two-loop recursion) on the north-star workload: extended
Rosenbrock, n = 1e8, m = 10, f64, LineSearchMoreThuente, inputs generated in HBM from a counter hash
(no host copy of any n-vector).  For N > 1 every rank solves its own independent problem of that size on
its own GPU (the path shards by independent minimisations; no data-path collective) and the aggregate
iterations/s is reported ("scaling": "weak"); RCCL is used only for the barriers / final gather.  Launched under
torchrun the ranks come from the environment; launched plainly with --gpus N > 1 the script starts its N ranks itself
(one process per GPU) and fails if the box has fewer than N devices -- it never reports fewer GPUs than it was asked for.
The timed window always starts with the history full (c = m), whatever --warmup says.  The default workload's line also
carries the batched mode of BASELINE.json's cfg5 (the mode that shards problem ids over the GPUs) as "cfg5_batched".
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (about 6.3 TB/s achievable)


def pmc_traffic(n, m, fused=False):
    """HBM bytes per two-loop step from a committed rocprofv3 PMC summary (profiles/*_pmc_summary.json, produced by
    scripts/profile.sh + scripts/summarize_profile.py on this same command) -- used only when that profile was taken
    at exactly this (n, m); PMC counters cannot be collected from inside the timed process."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json"))):
        try:
            d = json.load(open(f))
            t = d.get("twoloop_avg_hbm_bytes_per_launch")
            if t and int(d.get("n", 0)) == int(n) and int(d.get("m", 0)) == int(m) and bool(d.get("fused_post", False)) == fused:
                best = {"bytes_per_launch": t, "source": os.path.relpath(f, ROOT)}
        except Exception:
            pass
    return best


def cpu_baseline(args):
    """Reference (unmodified headers + eigen_shim, native accumulators) on ONE host core, bounded sample."""
    import numpy as np

    import oracle_lib as O
    fam, kind = ("ref", "reference") if O.available("ref", "native") else ("port", "port")
    if not O.available(fam, "native"):
        return None
    orc = O.Oracle(fam, "native")
    n = int(args.cpu_n)
    warm, timed = args.m + 2, args.cpu_steps
    x0 = O.rosen_x0(n)
    # two runs that differ only in max_iterations: the difference isolates `timed` steady-state iterations
    p1 = O.lbfgs_params(m=args.m, epsilon=0, epsilon_rel=0, max_iterations=warm)
    p2 = O.lbfgs_params(m=args.m, epsilon=0, epsilon_rel=0, max_iterations=warm + timed)
    t0 = time.perf_counter()
    _, r1 = orc.lbfgs(O.F64, O.LS_MT, O.OBJ_ROSEN, x0, p1)
    t1 = time.perf_counter()
    _, r2 = orc.lbfgs(O.F64, O.LS_MT, O.OBJ_ROSEN, x0, p2)
    t2 = time.perf_counter()
    dt = (t2 - t1) - (t1 - t0)
    its = (r2.niter - r1.niter) / dt
    scale = n / float(args.n)
    return {"value": its * scale, "unit": "iterations/s", "cores": 1, "kind": kind,
            "extrapolated": True, "measured_value": its, "measured_n": n,
            "sample": "same workload at n=%d (%.3g of n), %d timed iterations after %d warm-up, %.1f s of CPU; "
                      "measured %.4g it/s scaled linearly by n ratio; host has %d cores"
                      % (n, scale, r2.niter - r1.niter, r1.niter, t2 - t0, its, os.cpu_count())}


def cpu_baseline_full(args):
    """the 1-core reference at the metric's OWN size (n = 1e8, m = 10), un-extrapolated: m+2 warm-up iterations
    (history full) + a few timed ones in ONE run.  The oracle stamps every functor call (oracle_*_set_eval_clock); the gap
    between two calls that holds a two-loop recursion (~(8m+1) n elements) is an order of magnitude longer than the gap
    between two trials of one line search, which marks the iteration boundaries; the count of boundaries is checked against
    the iterations the run reports.  Skipped (None) when the host cannot hold the ~(2m+12) n doubles."""
    import numpy as np

    import oracle_lib as O
    fam, kind = ("ref", "reference") if O.available("ref", "native") else ("port", "port")
    if not O.available(fam, "native"):
        return None
    n, m = int(args.n), args.m
    need = (2 * m + 14) * n * 8
    try:
        avail = [int(ln.split()[1]) * 1024 for ln in open("/proc/meminfo") if ln.startswith("MemAvailable:")][0]
    except Exception:
        avail = 0
    if avail < 1.25 * need or (args.cpu_full == "auto" and avail < 48e9):
        return {"skipped": "host memory: %.0f GB available, %.0f GB needed" % (avail / 1e9, 1.25 * need / 1e9)}
    orc = O.Oracle(fam, "native")
    setc = getattr(orc.lib, ("oracle_ref" if fam == "ref" else "oracle_port") + "_set_eval_clock")
    setc.argtypes = [C.POINTER(C.c_double), C.c_int]
    warm, timed = m + 2, args.cpu_full_steps
    stamps = np.zeros(4096)
    x0 = O.rosen_x0(n)
    setc(stamps.ctypes.data_as(C.POINTER(C.c_double)), stamps.size)
    t0 = time.perf_counter()
    try:
        x_ref, r = orc.lbfgs(O.F64, O.LS_MT, O.OBJ_ROSEN, x0, O.lbfgs_params(m=m, epsilon=0, epsilon_rel=0, max_iterations=warm + timed))
    finally:
        setc(None, 0)
    t1 = time.perf_counter()
    del x0
    # the reference's iterate after these iterations is the oracle side of `parity` (north_star_parity): the GPU solver runs
    # the same instance for the same number of iterations
    args._ref_run = {"x": x_ref, "niter": r.niter, "nfev": r.nfev, "fx": r.fx, "n": n, "m": m, "kind": kind,
                     "oracle": "%s, native accumulators (plain f64 sums in index order), 1 core" %
                               ("reference headers + eigen_shim" if fam == "ref" else "restatement of the reference")}
    if args.cpu_full_dd and O.available(fam, "dd"):
        # the same run with the reference built on the parity contract's extended sums: not timed
        x_dd, r_dd = O.Oracle(fam, "dd").lbfgs(O.F64, O.LS_MT, O.OBJ_ROSEN, O.rosen_x0(n),
                                               O.lbfgs_params(m=m, epsilon=0, epsilon_rel=0, max_iterations=warm + timed))
        args._ref_run_dd = {"x": x_dd, "niter": r_dd.niter, "nfev": r_dd.nfev, "fx": r_dd.fx, "n": n, "m": m, "kind": kind,
                            "oracle": "%s, double-double accumulators (the parity contract's arithmetic), 1 core" %
                                      ("reference headers + eigen_shim" if fam == "ref" else "restatement of the reference")}
    ts = stamps[:min(r.nfev, stamps.size)]
    gaps = np.diff(ts)
    thr = float(np.sqrt(gaps.min() * gaps.max()))  # geometric middle of "next trial" and "next iteration" gaps
    # last functor call of every iteration; the gap after call 0 (f at x0; first-touch page faults of the history) is no
    # boundary: the first line search starts from it directly (LBFGS.h:91-108)
    ends = [i for i, g in enumerate(gaps) if g > thr and i > 0] + [len(ts) - 1]
    ok = len(ends) == r.niter and r.niter == warm + timed and gaps.max() > 3 * gaps.min()
    if ok:
        dt = float(ts[ends[-1]] - ts[ends[-1 - timed]])
        its, how = timed / dt, ("iterations %d..%d of one run, boundaries from the functor-call clock (%d boundaries = %d "
                                "iterations)" % (warm + 1, warm + timed, len(ends), r.niter))
    else:  # could not separate the iterations: whole-run average, first (short-history) iterations included
        its, how = r.niter / (t1 - t0), "whole run of %d iterations (boundaries not separable: %d found)" % (r.niter, len(ends))
    return {"value": its, "unit": "iterations/s", "cores": 1, "kind": kind, "extrapolated": False, "measured_n": n,
            "timed_iterations": timed if ok else r.niter, "nfev": r.nfev,
            "sample": "the metric's own size n=%d, m=%d: %s; %.1f s of CPU in all; host has %d cores"
                      % (n, m, how, t1 - t0, os.cpu_count())}


def north_star_parity(args, local, ref):
    """north_star: "iterate trajectory matching the CPU reference to 1e-10" on the benchmark's OWN instance (n = 1e8, m = 10,
    extended Rosenbrock from the counter-hash x0, More-Thuente).  `ref` is what cpu_baseline_full kept of the reference's run
    (the iterate after K iterations, its evaluation count and objective value); the product path runs the same instance for K
    iterations here and every coordinate is compared.  The reference side uses its native accumulators, so this is the
    un-helped comparison: the GPU path (double-double sums) follows it to <= 1e-10 for about 30 iterations (README); with
    --cpu-full-dd the reference is built with the same extended sums and the expectation is 0."""
    import numpy as np

    import lbfgspp_amd as A
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    n, m, K = ref["n"], ref["m"], ref["niter"]
    s = A.LBFGSSolver(A.LBFGSParam(m=m, epsilon=0.0, epsilon_rel=0.0, past=0, max_iterations=K), linesearch=A.LS_MORE_THUENTE,
                      dtype="float64", device=local)
    ctx = s.prepare(n)
    L.check(core.lbfgsx_gen_rosen_x0(ctx, 7))
    L.check(core.lbfgsx_sync(ctx))
    niter, fx = s.minimize_resident(A.ExtendedRosenbrock(), n)
    x = np.empty(n)
    L.check(core.lbfgsx_download(ctx, L.VEC_X, x.ctypes.data_as(C.c_void_p)))
    nfev = s.last.nfev
    s.close()
    xr = ref["x"]
    # all n coordinates, in slabs (no second 800 MB temporary)
    worst, at = 0.0, 0
    for lo in range(0, n, 1 << 24):
        d = np.abs(x[lo:lo + (1 << 24)] - xr[lo:lo + (1 << 24)])
        k = int(d.argmax())
        if d[k] > worst:
            worst, at = float(d[k]), lo + k
    stride = max(1, n // 4096)
    return {"n": n, "m": m, "iterations": int(niter), "iterations_equal": bool(niter == ref["niter"]),
            "nfev": int(nfev), "nfev_equal": bool(nfev == ref["nfev"]),
            "max_abs_dx": worst, "max_abs_dx_at": at, "max_abs_dx_strided_sample": float(np.abs(x[::stride] - xr[::stride]).max()),
            "x_inf_norm": float(np.abs(xr[::stride]).max()),
            "fx": fx, "fx_oracle": ref["fx"], "fx_rel_diff": abs(fx - ref["fx"]) / max(abs(ref["fx"]), 1e-300),
            "tolerance": 1e-10, "within_tolerance": bool(worst <= 1e-10 and nfev == ref["nfev"] and niter == ref["niter"]),
            "oracle": ref["oracle"],
            "instance": "the benchmark's own: extended Rosenbrock n=%d, x0 = -1.2 / 1 + 0.4 u01(counter hash, seed 7), m=%d, "
                        "LineSearchMoreThuente, %d iterations from x0" % (n, m, K)}


def cpu_baseline_all_cores(args):
    """SURVEY.md 8(d)'s stronger CPU point: the restatement with its n-length loops under OpenMP (native accumulators,
    oracle/liboracle_native_omp.so) on every host core.  Reported next to `cpu_baseline`, never instead of it."""
    import oracle_lib as O
    if not O.available("port", "native_omp"):
        return None
    orc = O.Oracle("port", "native_omp")
    # the cores this process may actually use: the cgroup CPU quota when there is one (the GPU boxes expose 256 logical
    # CPUs under a 16-CPU quota; 256 threads on 16 CPUs run 10x slower than 16)
    cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(round(float(quota) / float(period)))))
    except Exception:
        pass
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    f = orc.lib.oracle_port_set_threads
    f.argtypes = [C.c_int]
    f.restype = C.c_int
    cores = f(int(cores))
    n = int(args.cpu_n_all)
    warm, timed = args.m + 2, 8
    x0 = O.rosen_x0(n)
    p1 = O.lbfgs_params(m=args.m, epsilon=0, epsilon_rel=0, max_iterations=warm)
    p2 = O.lbfgs_params(m=args.m, epsilon=0, epsilon_rel=0, max_iterations=warm + timed)
    orc.lbfgs(O.F64, O.LS_MT, O.OBJ_ROSEN, x0, O.lbfgs_params(m=args.m, epsilon=0, epsilon_rel=0, max_iterations=2))  # page in
    t0 = time.perf_counter()
    _, r1 = orc.lbfgs(O.F64, O.LS_MT, O.OBJ_ROSEN, x0, p1)
    t1 = time.perf_counter()
    _, r2 = orc.lbfgs(O.F64, O.LS_MT, O.OBJ_ROSEN, x0, p2)
    t2 = time.perf_counter()
    dt = (t2 - t1) - (t1 - t0)
    its = (r2.niter - r1.niter) / dt
    scale = n / float(args.n)
    return {"value": its * scale, "unit": "iterations/s", "cores": cores, "kind": "port",
            "extrapolated": True, "measured_value": its, "measured_n": n,
            "sample": "restatement under OpenMP (every core of the CPU quota, native accumulators) at n=%d "
                      "(%.3g of n), %d timed iterations after %d warm-up, %.1f s of CPU wall; measured %.4g it/s scaled "
                      "linearly by n ratio" % (n, scale, r2.niter - r1.niter, r1.niter, t2 - t0, its)}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, with the
    environment torchrun would give them.  Rank 0's stdout is the parent's (the one JSON line)."""
    import socket
    import subprocess
    import lbfgspp_amd as A
    core, _ = A.load()
    ndev = core.lbfgsx_device_count()
    forced = os.environ.get("LBFGSX_BENCH_FORCE_DEVICE")
    if ndev < args.gpus and forced is None:
        raise SystemExit("bench.py --gpus %d: this box has %d GPU(s); refusing to report fewer GPUs than asked for "
                         "(LBFGSX_BENCH_FORCE_DEVICE=k shares device k between the ranks, protocol test only)"
                         % (args.gpus, ndev))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if forced is not None:
            env.setdefault("LBFGSX_PERSIST", "0")  # ranks sharing one device cannot each own all of its CUs
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    raise SystemExit(rc)


def init_dist(args):
    """One process per GPU (torchrun env, or spawn_ranks).  Backend nccl (= RCCL over xGMI); LBFGSX_BENCH_FORCE_DEVICE=k
    (with gloo) exists only to exercise the N > 1 code path on a single-GPU box."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    forced = os.environ.get("LBFGSX_BENCH_FORCE_DEVICE")
    dev = int(forced) if forced is not None else local
    backend = os.environ.get("LBFGSX_BENCH_BACKEND", "gloo" if forced is not None else "nccl")
    comm_dev = torch.device("cpu")
    if world > 1:
        if backend == "nccl":
            torch.cuda.set_device(dev)
            comm_dev = torch.device("cuda", dev)
            dist.init_process_group("nccl", device_id=comm_dev)
        else:
            dist.init_process_group(backend)
    return rank, world, dev, comm_dev, dist


def gather_objects(obj, rank, world, dist):
    """Every rank's small record, in rank order, on every rank (the list is only reported by rank 0)."""
    if world == 1:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def collective_info(dist, world):
    """Which collective library carried the barriers / reductions / gathers of an N > 1 run."""
    import torch
    info = {"backend": dist.get_backend(), "world_size": world, "rccl_version": None,
            "data_path_collectives": 0,
            "used_for": "timing barriers, max-over-ranks of the elapsed time, gather of the per-rank / per-problem records"}
    try:
        if info["backend"] == "nccl":
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        pass
    return info


def run_batched(args, rank, world, local, comm_dev, dist, steps):
    """BASELINE.json cfg5: independent extended-Rosenbrock problems n=1e5, m=10, f32, LineSearchMoreThuente, fixed
    budget of `steps` iterations per problem; every rank solves its own contiguous shard of problem ids in the
    lock-step batch (no data-path collective) and the per-problem records are all-gathered at the end.
    Returns the result object on rank 0 (None elsewhere)."""
    import numpy as np
    import torch

    import lbfgspp_amd as A
    from lbfgspp_amd import batched as B
    n, m, P = int(args.batched_n), 10, args.problems_per_gpu
    total = P * world
    first, count = B.shard_range(total, rank, world)
    par = A.LBFGSParam(m=m, epsilon=0.0, epsilon_rel=0.0, max_iterations=steps)
    # Setup, outside the timed window: the resident batch (~(2m+9) n P floats = 11.9 GB for P = 1024) is allocated ONCE and
    # kept across minimisations (lbfgsx_lockstep_create); then full-size warm-up solves of the very instance that is timed.
    # Inputs (the start points) are generated in HBM by the solve itself, as in every other leg.
    ts = time.perf_counter()
    bdt = np.float64 if args.batched_dtype == "f64" else np.float32
    esz = 8.0 if args.batched_dtype == "f64" else 4.0
    batch = B.LockstepBatch(par, n, count, dtype=bdt, device=local)
    for _ in range(max(1, min(args.warmup, 2))):
        batch.minimize(first=first, seed_base=1000)
    setup_s = time.perf_counter() - ts

    def barrier():
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    recs = batch.minimize(first=first, seed_base=1000)
    barrier()
    elapsed = local_elapsed = time.perf_counter() - t0
    st = dict(batch.stats)
    # the same solve once more with a pair of events around every launch (untimed): the kernels' share of a lock-step iteration
    batch.set_timing(True)
    recs2 = batch.minimize(first=first, seed_base=1000)
    kst = dict(batch.stats)
    batch.close()
    if not (np.array_equal(recs["niter"], recs2["niter"]) and np.array_equal(recs["fx"], recs2["fx"])):
        raise SystemExit("bench.py: two minimisations of one resident batch disagree")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    full = B.gather_records(recs, total, rank, world, dist=dist if world > 1 else None,
                            device=comm_dev if world > 1 else None)
    per_rank = gather_objects({"rank": rank, "device": local, "first_problem": int(first), "problems": int(count),
                               "seconds": local_elapsed, "value": float(recs["niter"].sum()) / local_elapsed}, rank, world, dist)
    if rank != 0:
        return None
    its, fev = int(full["niter"].sum()), int(full["nfev"].sum())
    # SURVEY 8(d) algorithmic bytes: (8m+12) n per iteration + 4n per extra trial
    alg_ = (its * (8 * m + 12) + (fev - its) * 4) * n * esz
    # HBM traffic model of the one-launch lock-step iteration (batched_iter.hip; the direction stays on the CU): per problem and
    # iteration k the post pass 6 n, the recursion over c = min(k-1, m) pairs (4c+2) n, drt + the first trial 4 n; 4 n per
    # further trial; 2 n for the evaluation at x0
    hbm_ = 0.0
    for it_, fe_ in zip(full["niter"], full["nfev"]):
        it_, fe_ = int(it_), int(fe_)
        hbm_ += sum(6 + 4 * min(k - 1, m) + 2 + 4 for k in range(1, it_ + 1)) + 4 * max(fe_ - 1 - it_, 0) + 2
    hbm_ *= n * esz
    model_gbs = hbm_ / elapsed / 1e9 / world
    return {
        "metric": "batched L-BFGS problem-iterations/sec (cfg5: n=%g, m=10, f32)" % n, "value": its / elapsed,
        "unit": "problem-iterations/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": elapsed / max(steps, 1) * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.batched_dtype, "data": "synthetic",
        "config": {"workload": ("cfg5: %d independent extended-Rosenbrock problems per GPU, n=%g, m=10, {DT}, "
                                "LineSearchMoreThuente, %d iterations each, lock-step batch; contiguous problem-id blocks "
                                "per rank, no data-path collective, one all-gather of the result records"
                                % (P, n, steps)).replace("{DT}", args.batched_dtype),
                   "problems_total": total, "fevals_total": fev, "failed": int((full["status"] != 0).sum()),
                   "one_launch_per_iteration": bool(st.get("fused")), "lockstep_iterations": st.get("lockstep_iterations"),
                   "setup_seconds": setup_s,
                   "setup": "resident batch allocated once + %d full-size warm-up solve(s) of the timed instance, outside the "
                            "timed window; the timed window is one minimisation of all problems from their start points "
                            "(start points generated in HBM inside it)" % max(1, min(args.warmup, 2)),
                   # host share of a lock-step iteration: wall (timed run) against the kernels' own durations (events around
                   # every launch of an identical, untimed run)
                   "wall_ms_per_step": local_elapsed / max(steps, 1) * 1e3,
                   "kernel_ms_per_step": kst["kernel_ms"] / max(steps, 1),
                   "launches_per_step": kst["launches"] / float(max(steps, 1)),
                   "host_waits_per_step": st["waits"] / float(max(steps, 1)), "wait_timeouts": st["wait_timeouts"],
                   "per_rank": per_rank if world > 1 else None},
        "roofline": dict({"bound": "hbm", "achieved": model_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": model_gbs / HBM_PEAK_GBS,
                     "hbm_model_bytes_per_problem_iteration": hbm_ / max(its, 1),
                     "algorithmic_GBs": alg_ / elapsed / 1e9 / world,
                     "note": "end to end per GPU, host control flow included. achieved = HBM traffic model of the "
                             "one-launch lock-step iteration ((4c+12) n elements per iteration: the direction stays on the CU) / wall time; "
                             "algorithmic_GBs = SURVEY 8(d)'s (8m+12) n per iteration / wall time, which counts the q "
                             "traffic that never reaches HBM and may therefore exceed the peak"},
                     **traffic_fields(leg_traffic("cfg5", n, m) if args.batched_dtype == "f32" else None))}


def leg_traffic(kind, n, m):
    """HBM bytes of one leg from a committed rocprofv3 PMC summary (profiles/*_legs_pmc_summary.json, written by
    scripts/r4/pmc_legs.py from FETCH_SIZE / WRITE_SIZE passes of the leg's own command) -- static, labelled, and only
    when the summary was taken at exactly this (kind, n, m): counters cannot be read inside the timed process."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_legs_pmc_summary.json"))):
        try:
            for e in json.load(open(f)).get("legs", []):
                if e.get("leg") == kind and int(e.get("n", 0)) == int(n) and int(e.get("m", 0)) == int(m) and e.get("hbm_bytes"):
                    best = dict(e, source=os.path.relpath(f, ROOT))
        except Exception:
            pass
    return best


def traffic_fields(t):
    """The `traffic*` keys of a leg's roofline object from leg_traffic()'s entry (None: all null)."""
    if not t:
        return {"traffic": None, "traffic_static": None, "traffic_source": None}
    return {"traffic": t["hbm_bytes"], "traffic_static": True, "traffic_per": t.get("per"),
            "traffic_source": t["source"] + " (" + t.get("how", "rocprofv3 PMC passes of this leg's command, taken separately") + ")"}


def run_cfg4(args, rank, world, local, comm_dev, dist, iters=40, m=10):
    """BASELINE.json cfg4: L-BFGS-B (generalized Cauchy point + BOXCQP subspace minimisation) on the box-constrained diag
    quadratic, n=1e7, m=10 (or another history length: the `cfg4_m20` leg), lb=-1, ub=1, x0=0, f64 -- `iters` iterations from
    x0 on every rank (independent problems, seed 1+rank).  Two figures, each with BOTH sides of its roofline fraction taken
    from the same iterations: the steady state = the second half of the run (mean iteration time of that window; the sweeps q,
    the sorted break points and the launches / host synchronisations / copies counted by the library are snapshotted at the
    iteration hook, so they are that window's, too), and the run from x0 (everything over all iterations).  Returns the object
    on rank 0."""
    import numpy as np
    import torch  # noqa: F401

    import lbfgspp_amd as A
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    n = int(args.cfg4_n)

    def setup(solver, nn, seed):
        ctx = solver.prepare(nn)
        L.check(core.lbfgsx_gen_diag_quad(ctx, 10.0, seed))
        L.check(core.lbfgsx_fill(ctx, L.VEC_X, 0.0))
        L.check(core.lbfgsx_fill(ctx, L.VEC_LB, -1.0))
        L.check(core.lbfgsx_fill(ctx, L.VEC_UB, 1.0))
        L.check(core.lbfgsx_sync(ctx))
        return ctx
    # untimed small solve of the same problem: HIP loads a kernel's code object at its first launch (~60 kernels)
    w = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=m + 4), device=local)
    setup(w, 1 << 18, 1)
    w.minimize_resident(A.DiagQuadratic(), 1 << 18)
    w.close()
    s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters), device=local)
    setup(s, n, 1 + rank)

    def barrier():
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
    cnt = (C.c_int64 * 8)()
    snaps = []  # (time, solver statistics, library counters) at the end of every iteration

    def hook(k):
        t = time.perf_counter()
        core.lbfgsx_counters_ex(C.byref(cnt), 0)
        snaps.append((t, s.stats(), tuple(cnt[i] for i in range(6))))
    s.set_iteration_hook(hook)
    barrier()
    core.lbfgsx_counters_ex(None, 1)
    t0 = time.perf_counter()
    niter, fx = s.minimize_resident(A.DiagQuadratic(), n)
    t1 = time.perf_counter()
    core.lbfgsx_counters_ex(C.byref(cnt), 0)
    barrier()
    st = s.stats()
    nfev = s.last.nfev
    s.close()
    stamps = [v[0] for v in snaps]
    per = np.diff(np.array([t0] + stamps))
    # steady window: iterations w0+1 .. len(snaps) (the hook does not fire after the last iteration, which ends the run)
    if len(snaps) < 2:
        raise SystemExit("bench.py: the cfg4 leg needs at least 3 iterations (--cfg4-iters)")
    w0 = max(1, len(snaps) // 2)
    win = per[w0:]
    total, steady_ms, steady_med_ms = t1 - t0, float(win.mean()) * 1e3, float(np.median(win)) * 1e3
    first_ms = float(per[0]) * 1e3
    if world > 1:
        t = torch.tensor([total, steady_ms, steady_med_ms], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        total, steady_ms, steady_med_ms = float(t[0].item()), float(t[1].item()), float(t[2].item())
    if rank != 0:
        return None
    a_st, b_st = snaps[w0 - 1][1], snaps[-1][1]
    a_c, b_c = snaps[w0 - 1][2], snaps[-1][2]
    nwin = len(win)

    def d(key):
        return b_st[key] - a_st[key]

    def bytes_per_iteration(q, n_sorted):
        # algorithmic bytes of an L-BFGS-B iteration with q BOXCQP sweeps and one objective evaluation,
        # [(4m + 19) + (q + 1)(4m + 1)] n elements (history full); radix-sort traffic (~96 B per sorted break point) on top
        return ((4 * m + 19) + (q + 1.0) * (4 * m + 1)) * n * 8 + 96.0 * n_sorted
    # the window's own counts
    q_w = d("submin_sweeps") / max(1, d("submin_calls"))
    searches_w = max(1, d("gcp_searches"))
    n_ord_w, n_sorted_w = d("gcp_nord") / searches_w, d("gcp_sorted") / searches_w
    bytes_w = bytes_per_iteration(q_w, n_sorted_w)
    # ... and the whole run's, for the from-x0 figure
    q = st["submin_sweeps"] / max(1, st["submin_calls"])
    searches = max(1, st["gcp_searches"])
    n_ord, n_sorted = st["gcp_nord"] / searches, st["gcp_sorted"] / searches
    bytes_it = bytes_per_iteration(q, n_sorted)
    steady = 1e3 / steady_ms
    # The byte model of the path AS BUILT (lbfgsx_counters_ex): every launch adds what its pass has to
    # move for the rows and columns it was launched over -- the compact copy holds the free rows only, a carried Gram and a
    # sweep's W_P'rhs from held sums need no pass, so this is well below the reference's statement count above.  The
    # roofline fraction is model bytes of the window / wall time of the window / peak: at most 1 by construction.
    model_w = (b_c[3] - a_c[3]) / max(1, nwin)
    model_x0 = cnt[3] / max(1, niter)
    passes_w = (b_c[4] - a_c[4]) / max(1, nwin)
    n_free_w = (b_c[5] - a_c[5]) / max(1, b_c[4] - a_c[4])
    ach_steady, ach_x0 = model_w * steady / 1e9, model_x0 * (niter / total) / 1e9
    ref_steady, ref_x0 = bytes_w * steady / 1e9, bytes_it * (niter / total) / 1e9
    tr = leg_traffic("cfg4", n, m)
    return {
        "metric": "L-BFGS-B iterations/sec at n=%d, m=%d (box-constrained diag quadratic); steady state" % (n, m),
        "value": world * steady, "unit": "iterations/s", "n_gpus": world, "steps": iters,
        "ms_per_step": steady_ms, "ms_per_step_median": steady_med_ms, "value_median": world * 1e3 / steady_med_ms,
        "scaling": "weak", "dtype": "f64",
        "from_x0": {"value": world * niter / total, "unit": "iterations/s", "iterations": niter, "seconds": total,
                    "first_iteration_ms": first_ms,
                    "q": q, "n_ord": n_ord, "n_sorted": n_sorted, "gcp_crossings": st["gcp_crossings"] / max(1, niter),
                    "launches_per_iteration": cnt[0] / max(1, niter), "host_syncs_per_iteration": cnt[1] / max(1, niter),
                    "copies_per_iteration": cnt[2] / max(1, niter),
                    "note": "all %d iterations from x0 = 0, the first Cauchy searches (millions of crossings) included" % niter},
        "config": {"workload": "cfg4: L-BFGS-B, f = 0.5 ||diag(a) x - b||^2, kappa=10, lb=-1, ub=1, x0=0, n=%d, m=%d, f64, "
                               "LineSearchMoreThuente, %d iterations from x0 (epsilon=epsilon_rel=0, past=0); value = steady "
                               "state = 1 / MEAN per-iteration wall time of the second half of the run (value_median beside "
                               "it); q, n_ord, n_sorted, crossings, launches / synchronisations / copies below are those of "
                               "the SAME iterations" % (n, m, iters),
                   "n": n, "m": m, "iterations": niter, "fevals_total": nfev, "fx": fx,
                   "window": {"first_iteration": w0 + 1, "last_iteration": w0 + nwin, "iterations": nwin,
                              "history_full": bool(w0 >= m)},
                   "q": q_w, "n_ord": n_ord_w, "n_sorted": n_sorted_w, "gcp_crossings": d("gcp_crossings") / max(1, nwin),
                   "gcp_crossings_total": st["gcp_crossings"], "gcp_dev_crossings_total": st["gcp_dev_crossings"],
                   "submin_calls": d("submin_calls"), "submin_sweeps": d("submin_sweeps"), "gram_carried": d("gram_carried"),
                   "submin_calls_total": st["submin_calls"], "submin_sweeps_total": st["submin_sweeps"],
                   "gram_carried_total": st["gram_carried"], "rhs_identities": d("rhs_identities"),
                   "rhs_identities_total": st.get("rhs_identities"),
                   "n_free": n_free_w, "compact_passes_per_iteration": passes_w,
                   "launches_per_iteration": (b_c[0] - a_c[0]) / max(1, nwin),
                   "host_syncs_per_iteration": (b_c[1] - a_c[1]) / max(1, nwin),
                   "copies_per_iteration": (b_c[2] - a_c[2]) / max(1, nwin),
                   "per_iteration_ms": [round(float(v) * 1e3, 3) for v in per],
                   "phase_ms_per_iteration": {"cauchy": d("gcp_total_us") / 1e3 / max(1, nwin), "subspace": d("submin_us") / 1e3 / max(1, nwin),
                                              "linesearch": d("linesearch_us") / 1e3 / max(1, nwin)}},
        "roofline": dict({"bound": "hbm", "kernel": "whole L-BFGS-B iteration (passes over the compact copy of the free rows dominate; no "
                                               "single kernel holds more than a fifth of the time)",
                     "achieved": ach_steady, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_steady / HBM_PEAK_GBS,
                     "achieved_from_x0": ach_x0, "frac_from_x0": ach_x0 / HBM_PEAK_GBS,
                     "model_bytes": model_w, "model_bytes_from_x0": model_x0,
                     # SURVEY 8(d)'s count of the reference's statements, kept for comparison only: the path as built moves
                     # fewer bytes than that, so "statement bytes / time" is not a fraction of anything (it exceeds the peak)
                     "reference_statement_bytes": bytes_w, "reference_statement_bytes_from_x0": bytes_it,
                     "reference_statement_GBs": ref_steady, "reference_statement_GBs_from_x0": ref_x0,
                     # what the counters saw (static, from the committed PMC summary of this leg)
                     "traffic_GBs": (tr["hbm_bytes"] * steady / 1e9) if tr else None,
                     "traffic_frac": (tr["hbm_bytes"] * steady / 1e9 / HBM_PEAK_GBS) if tr else None,
                     "model_over_traffic": (model_w / tr["hbm_bytes"]) if tr else None,
                     "note": "achieved = bytes the launches of the window had to move (lbfgsx_counters_ex: columns x rows of the compact "
                             "copy + the vectors each pass reads / writes per row, gathers at 64 B sectors; DESIGN.md section 5) / wall "
                             "time of the same iterations (host control flow included); the steady figure from the second half "
                             "of the run, the from-x0 figure from the whole run; no full Gram pass falls into the steady "
                             "window (the carried Gram refreshes every 256 iterations)"},
                     **traffic_fields(tr))}


def lbfgs_leg(args, rank, world, local, comm_dev, dist, **over):
    """One more L-BFGS configuration of BASELINE.json through the same code as the headline (run_north_star), reduced to
    the keys the line carries per leg."""
    import copy
    a = copy.copy(args)
    for k, v in over.items():
        setattr(a, k, v)
    a.workload, a.recursion = "north-star", "vector"
    out = run_north_star(a, rank, world, local, comm_dev, dist)
    if rank != 0:
        return None
    return {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype",
                                "config", "roofline")}


def run_north_star(args, rank, world, local, comm_dev, dist):
    import torch  # noqa: F401  (first: its bundled HIP runtime must be the process-wide one)

    import lbfgspp_amd as A
    from lbfgspp_amd import _lib as L

    core, _ = A.load()
    if core.lbfgsx_device_count() < 1:
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")

    sharded = args.workload == "sharded"
    if sharded and args.recursion == "vector":
        args.recursion = "gram"  # the only form whose reductions can cross devices
    n, m, K = int(args.n), args.m, args.steps
    # the timed window starts with the history full (c = m).  Iteration k's apply_Hv sees min(k, m) pairs,
    # so at least m untimed iterations come first whatever --warmup says; "warmup" in the line is what was asked for,
    # "warmup_run" what ran.
    W = max(args.warmup, m)
    n_global, shard_lo = n, 0
    if sharded:
        # contiguous row blocks, boundaries on multiples of 4 (whole Rosenbrock pairs, whole 16-byte vectors)
        per = (n_global // world) // 4 * 4
        shard_lo = rank * per
        n = (n_global - shard_lo) if rank == world - 1 else per
    ls = A.LS_MORE_THUENTE if args.objective == "rosenbrock" else A.LS_NOCEDAL_WRIGHT
    par = A.LBFGSParam(m=m, epsilon=0.0, epsilon_rel=0.0, past=0, max_iterations=W + K + 1)
    solver = A.LBFGSSolver(par, linesearch=ls, dtype="float64", device=local)
    gram = args.recursion != "vector"
    f32h = args.recursion == "gram-f32h"
    if gram:
        solver.set_recursion(L.RECURSION_GRAM_SPACE_F32H if f32h else L.RECURSION_GRAM_SPACE)
    ctx = solver.prepare(n)
    if sharded:
        L.check(core.lbfgsx_set_shard(ctx, shard_lo, n_global))
        # The driver's sums cross the shards through the LIBRARY's all-reduce (lbfgsx_comm_*, csrc/rccl_allreduce.hip: one
        # ncclAllReduce over xGMI per bundle), called from C without a Python frame in between.  The launcher's collective
        # library only distributes the communicator's id.  (Ranks that share one device -- the protocol test on a
        # one-GPU box, LBFGSX_BENCH_FORCE_DEVICE -- cannot be RCCL ranks; they fall back to the launcher's all-reduce.)
        n_reduces = [0]
        comm = C.c_void_p()
        comm_owner = True
        threads = isinstance(dist, ThreadDist)
        shared_device = os.environ.get("LBFGSX_BENCH_FORCE_DEVICE") is not None and world > 1
        if shared_device:
            def allreduce(v):
                n_reduces[0] += 1
                dist.all_reduce(torch.from_numpy(v))  # in place on the callback's memory
            solver.set_reducer(allreduce)
            comm = None
        elif threads:
            # one process drives every device: rank 0 forms the communicator over the device list, the others take it
            devs = gather_objects(local, rank, world, dist)
            handle = [None]
            if rank == 0:
                arr = (C.c_int * world)(*devs)
                L.check(core.lbfgsx_comm_create_local(C.byref(comm), arr, world))
                handle[0] = comm.value
            handle = gather_objects(handle[0], rank, world, dist)
            comm = C.c_void_p(handle[0])
            comm_owner = rank == 0
            solver.set_native_reducer(comm, rank)
        else:
            ident = [None]
            if rank == 0:
                buf = C.create_string_buffer(128)
                L.check(core.lbfgsx_comm_unique_id(buf))
                ident[0] = buf.raw
            if world > 1:
                dist.broadcast_object_list(ident, src=0)
            L.check(core.lbfgsx_comm_create_rank(C.byref(comm), local, rank, world, ident[0]))
            solver.set_native_reducer(comm, 0)
    if args.objective == "rosenbrock":
        L.check(core.lbfgsx_gen_rosen_x0(ctx, 7 + (0 if sharded else rank)))
        f = A.ExtendedRosenbrock()
    else:
        L.check(core.lbfgsx_gen_diag_quad(ctx, 10.0, 1 + (0 if sharded else rank)))
        L.check(core.lbfgsx_fill(ctx, L.VEC_X, 0.0))
        f = A.DiagQuadratic()
    L.check(core.lbfgsx_sync(ctx))

    def barrier():
        L.check(core.lbfgsx_sync(ctx))
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    marks = {}

    def hook(k):
        if k == W:
            marks["ncorr0"] = core.lbfgsx_bfgs_ncorr(ctx)
            barrier()
            L.check(core.lbfgsx_timing_enable(ctx, 2 if os.environ.get("LBFGSX_PERSIST") == "0" else 1))
            marks["t0"] = time.perf_counter()
        elif k == W + K:
            barrier()
            marks["t1"] = time.perf_counter()
            tl_ms, tl_n, hv_ms, hv_n = C.c_double(), C.c_int64(), C.c_double(), C.c_int64()
            L.check(core.lbfgsx_timing_read(ctx, C.byref(tl_ms), C.byref(tl_n), C.byref(hv_ms), C.byref(hv_n)))
            marks["tl"] = (tl_ms.value, tl_n.value, hv_ms.value, hv_n.value)
            core.lbfgsx_timing_fused_launches.restype = C.c_int64
            core.lbfgsx_timing_fused_launches.argtypes = [C.c_void_p]
            marks["fused"] = core.lbfgsx_timing_fused_launches(ctx)
            L.check(core.lbfgsx_timing_enable(ctx, 0))

    solver.set_iteration_hook(hook)
    niter, fx = solver.minimize_resident(f, n)
    if "t1" not in marks:
        raise SystemExit("solver stopped after %d iterations, before the timed window ended" % niter)
    elapsed = local_elapsed = marks["t1"] - marks["t0"]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    copy_gbs, triad_gbs = C.c_double(), C.c_double()
    L.check(core.lbfgsx_stream_probe(ctx, 10, C.byref(copy_gbs), C.byref(triad_gbs)))
    core.lbfgsx_persistent_launches.restype = C.c_int64
    core.lbfgsx_persistent_launches.argtypes = [C.c_void_p]
    persist_launches = core.lbfgsx_persistent_launches(ctx)
    core.lbfgsx_persistent_resident_elems.restype = C.c_int64
    core.lbfgsx_persistent_resident_elems.argtypes = [C.c_void_p]
    resident = core.lbfgsx_persistent_resident_elems(ctx) if persist_launches > 0 else 0
    core.lbfgsx_timing_fused_launches.restype = C.c_int64
    core.lbfgsx_timing_fused_launches.argtypes = [C.c_void_p]
    fused = int(marks.get("fused", 0))
    nfev, last_niter = solver.last.nfev, niter
    comm_info = None
    if sharded and comm is not None:
        info = (C.c_int * 4)()
        core.lbfgsx_comm_info(comm, C.byref(info))
        n_reduces[0] = int(core.lbfgsx_comm_calls(comm, rank if isinstance(dist, ThreadDist) else 0))
        comm_info = {"transport": "RCCL (ncclAllReduce, f64 sum) inside liblbfgsx.so" if info[2] else "host memory (ranks share a device)",
                     "ranks": int(info[0]), "rccl_version_code": int(info[3])}
        if world > 1:
            dist.barrier()   # every rank is done with the communicator
        if comm_owner:
            core.lbfgsx_comm_destroy(comm)
    del solver  # release the device context (the batched leg and the profilers' atexit handlers come next)
    import gc
    gc.collect()
    # N > 1: what every rank saw, so that the aggregate explains itself (which device, its own rate, its copy probe)
    ranks_counted = None
    if world > 1:
        # the ranks the collective library itself counts: a sum of ones over the timing communicator (N on an N-GPU run --
        # what lets a SCALE line prove that N ranks took part, whatever the launcher claims)
        one = torch.ones(1, dtype=torch.float64, device=comm_dev)
        dist.all_reduce(one)
        ranks_counted = int(round(float(one.item())))
    per_rank = gather_objects({"rank": rank, "device": local, "value": K / local_elapsed, "ms_per_step": local_elapsed / K * 1e3,
                               "stream_copy_GBs": copy_gbs.value, "rows": n}, rank, world, dist)
    if rank != 0:
        return None

    tl_ms, tl_n, hv_ms, hv_n = marks["tl"]
    esz = 8
    hv_bytes = (8 * m + 1) * n * esz  # algorithmic bytes per apply_Hv call, history full
    # launches that carry K3 as their step 0 (lbfgsx_post_linesearch_spec): K3's 6n elements ride along and the first
    # step's reads of grad and the new s (2n) are gone: (8m+5) n per launch.  The per-step figure stays launch / (2m+1).
    fused_all = fused > 0 and fused == hv_n
    if fused_all:
        hv_bytes = (8 * m + 5) * n * esz
    per_launch_bytes = hv_bytes / float(2 * m + 1)
    avg_launch_s = (tl_ms / max(tl_n, 1)) * 1e-3
    alg_gbs = per_launch_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    history_full = (marks["ncorr0"] == m) and (gram or tl_n == K * (2 * m + 1))
    # HBM-traffic model of the step: the persistent launch keeps `resident` elements of q on the CUs, and that share
    # of q's 4m reads / writes per apply_Hv never reaches HBM.  For n <= resident (cfg2) the algorithmic figure would
    # exceed the HBM peak, so `achieved` / `frac` quote the model and the algorithmic figure rides along.
    model_bytes = per_launch_bytes - (4 * m) * min(resident, n) * esz / float(2 * m + 1)
    model_gbs = model_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    use_model = alg_gbs > HBM_PEAK_GBS
    achieved = model_gbs if use_model else alg_gbs
    if gram:
        # tl_* = k_gs_post launches ((2m+6) n elements), hv_* = k_gs_combine launches ((2m+2) n elements)
        post_bytes, comb_bytes = (2 * m + 6) * n * esz, (2 * m + 2) * n * esz
        if f32h:  # the 2m history columns (and the two written ones) are 4-byte elements
            post_bytes, comb_bytes = (2 * m * 4 + 4 * esz + 2 * 4) * n, (2 * m * 4 + 2 * esz) * n
        post_s, comb_s = tl_ms / max(tl_n, 1) * 1e-3, hv_ms / max(hv_n, 1) * 1e-3
        achieved = post_bytes / post_s / 1e9 if post_s > 0 else 0.0
    pmc = pmc_traffic(n, m, fused_all) if persist_launches > 0 and not gram else None
    out = {
        # BASELINE.json's metric string for the north-star configuration; other sizes say what they are
        "metric": ("L-BFGS iterations/sec of ONE problem n=%d row-sharded over %d GPU(s), m=%d (%s), Gram-space recursion%s "
                   "(opt-in, not the bit-parity path); achieved HBM GB/s per GPU vs peak"
                   % (n_global, world, m, args.objective, " with f32 history" if f32h else "")) if sharded else
                  ("L-BFGS iterations/sec at n=%d, m=%d (%s), Gram-space recursion%s (opt-in, not the bit-parity path); "
                   "achieved HBM GB/s vs peak" % (n, m, args.objective, " with f32 history" if f32h else "")) if gram else
                  ("L-BFGS iterations/sec at n=10^8, m=10; achieved HBM GB/s vs peak"
                   if (n == 100000000 and m == 10 and args.objective == "rosenbrock") else
                   "L-BFGS iterations/sec at n=%d, m=%d (%s); achieved HBM GB/s vs peak" % (n, m, args.objective)),
        "value": (1 if sharded else world) * K / elapsed,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": K,
        "warmup": args.warmup,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if sharded else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "north-star: extended Rosenbrock n=%d m=%d f64 LineSearchMoreThuente, x0 from counter hash"
                               % (n, m) if args.objective == "rosenbrock" else
                               "diag quadratic kappa=10 n=%d m=%d f64 LineSearchNocedalWright" % (n, m),
                   "n": n, "m": m, "problems_per_gpu": 1, "warmup_run": W, "history_full": bool(history_full),
                   "fevals_total": nfev, "iterations_total": last_niter,
                   "apply_Hv_persistent_launches": int(persist_launches), "fused_post_launches_timed": fused},
        "roofline": {"bound": "hbm",
                     "kernel": ("k_twoloop_persist<fused post> (one launch = K3's statements as step 0 + the 2c+1 two-loop "
                                "steps, (8m+5) n elements; the figures below are per step = launch / (2c+1))") if fused_all else
                               ("k_twoloop_persist (one launch per apply_Hv = 2c+1 two-loop steps, axpy + dot each; "
                                "the figures below are per step = launch / (2c+1))") if persist_launches > 0 else
                               "k_twoloop (two-loop recursion step: axpy + dot)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "achieved_is": ("hbm traffic model (q resident on the CUs is not counted); the algorithmic figure "
                                     "exceeds the HBM peak at this size" if use_model else "algorithmic bytes / time"),
                     "algorithmic_GBs": alg_gbs, "hbm_model_GBs": model_gbs,
                     "traffic": pmc["bytes_per_launch"] if pmc else None,
                     "traffic_static": True if pmc else None,
                     "traffic_source": (pmc["source"] + " (rocprofv3 PMC passes of this command at this n, m, taken "
                                                        "separately: counters cannot be read inside the timed run)") if pmc else None,
                     "algorithmic_bytes_per_launch": per_launch_bytes, "hbm_model_bytes_per_launch": model_bytes,
                     "q_resident_elems": int(min(resident, n)),
                     "avg_launch_ms": avg_launch_s * 1e3, "launches_timed": tl_n,
                     "apply_Hv_ms": hv_ms / max(hv_n, 1),
                     # bytes of the steps that actually ran in the window / their time
                     "apply_Hv_GBs": (per_launch_bytes * tl_n / (hv_ms * 1e-3) / 1e9) if hv_ms > 0 else None,
                     "stream_copy_GBs": copy_gbs.value, "stream_triad_GBs": triad_gbs.value,
                     "frac_of_stream_copy": achieved / copy_gbs.value if copy_gbs.value else None},
    }
    if world > 1:
        out["per_rank"] = per_rank
        out["collective"] = dict(collective_info(dist, world), ranks_counted_by_allreduce=ranks_counted,
                                 devices=[r["device"] for r in per_rank])
    if gram:
        out["config"]["recursion"] = "gram-space, f32 history" if f32h else "gram-space"
        if sharded:
            out["config"]["workload"] = ("one extended-Rosenbrock problem of n=%d rows, contiguous row blocks of %d over %d "
                                         "rank(s); %d all-reduces of <= %d doubles in the run"
                                         % (n_global, n, world, n_reduces[0], 6 * m + 7))
            out["config"]["n"] = n_global
            out["config"]["rows_per_gpu"] = n
            out["config"]["allreduce"] = comm_info or {"transport": "the launcher's all-reduce (ranks share one device)"}
        out["roofline"] = {"bound": "hbm", "kernel": ("k_gs_post_mx" if f32h else "k_gs_post") + " (s, y + Gram rows of the new pair and gradient, one pass)",
                           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": None, "algorithmic_bytes_per_launch": post_bytes, "avg_launch_ms": post_s * 1e3,
                           "launches_timed": tl_n,
                           "combine": {"kernel": "k_gs_combine_mx" if f32h else "k_gs_combine", "algorithmic_bytes_per_launch": comb_bytes,
                                       "avg_launch_ms": comb_s * 1e3,
                                       "achieved": comb_bytes / comb_s / 1e9 if comb_s > 0 else 0.0},
                           "stream_copy_GBs": copy_gbs.value, "stream_triad_GBs": triad_gbs.value}
    return out


class ThreadDist:
    """The few torch.distributed calls of this script over the threads of ONE process (--single-process: one host thread
    per GPU instead of one process per GPU)."""

    class ReduceOp:
        MAX, SUM = "max", "sum"

    def __init__(self, world):
        import threading
        self.world = world
        self._bar = threading.Barrier(world)
        self._slots = [None] * world
        self._tls = threading.local()

    def bind(self, rank):
        self._tls.rank = rank

    def barrier(self):
        self._bar.wait()

    def all_gather_object(self, out, obj):
        self._slots[self._tls.rank] = obj
        self._bar.wait()
        out[:] = list(self._slots)
        self._bar.wait()

    def all_reduce(self, t, op="sum"):
        parts = [None] * self.world
        self.all_gather_object(parts, t.clone())
        acc = parts[0].clone()
        for p_ in parts[1:]:
            acc = (acc + p_) if op == "sum" else acc.max(p_)  # rank order: identical on every thread
        t.copy_(acc)

    def get_backend(self):
        return "threads"


def run_single_process(args, ndev_asked):
    """--single-process --gpus N: ONE process drives all N GPUs -- one host thread and one solver context per device for the
    single-problem legs, one resident lock-step batch per device (lbfgsx_lockstep_*) for the batch, and the batch's one exchange step natively over
    RCCL (lbfgsx_rccl_allgather_records: ncclCommInitAll + a grouped ncclAllGather).  LBFGSX_BENCH_DEVICES=0,0 lists the
    devices explicitly (a device twice: protocol test on a one-GPU box)."""
    import threading

    import numpy as np
    import torch

    import lbfgspp_amd as A
    from lbfgspp_amd import batched as B
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    ndev = core.lbfgsx_device_count()
    env = os.environ.get("LBFGSX_BENCH_DEVICES")
    devices = [int(v) for v in env.split(",")] if env else list(range(ndev_asked))
    if len(devices) != ndev_asked or any(d < 0 or d >= ndev for d in devices):
        raise SystemExit("bench.py --single-process --gpus %d: devices %r on a box with %d GPU(s); refusing to report fewer "
                         "GPUs than asked for" % (ndev_asked, devices, ndev))
    world = ndev_asked
    td = ThreadDist(world)
    results, errors = [None] * world, []
    cpu = torch.device("cpu")

    def worker(r):
        td.bind(r)
        try:
            results[r] = run_all(args, r, world, devices[r], cpu, td, batched=False)
        except BaseException as e:  # noqa: BLE001  (a dead thread must not leave the others at a barrier)
            errors.append(e)
            td._bar.abort()
    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errors:
        raise SystemExit("bench.py --single-process: %r" % (errors[0],))
    out = results[0]
    out["config"]["process_model"] = "one process, one host thread + one solver context per GPU (devices %r)" % devices
    out["collective"] = {"backend": "threads for the barriers; RCCL (dlopen, ncclCommInitAll) for the batch's record gather",
                         "world_size": world, "data_path_collectives": 0}
    if args.workload == "north-star" and args.recursion == "vector" and not args.no_batched:
        n, m, P, steps = 100000, 10, args.problems_per_gpu, args.batched_steps
        total = P * world
        par = A.LBFGSParam(m=m, epsilon=0.0, epsilon_rel=0.0, max_iterations=steps)
        # one resident batch per listed device (allocated and warmed up outside the timed window, as in the one-process-per-GPU
        # form), each driven by its own host thread over its contiguous block of problem ids
        import threading
        shards = [B.shard_range(total, r, world) for r in range(world)]
        batches = [B.LockstepBatch(par, n, cnt, dtype=np.float32, device=devices[r]) for r, (_, cnt) in enumerate(shards)]
        parts = [None] * world

        def solve_all():
            errs = []

            def one(r):
                try:
                    parts[r] = batches[r].minimize(first=shards[r][0], seed_base=1000)
                except BaseException as e:  # noqa: BLE001
                    errs.append(e)
            ths = [threading.Thread(target=one, args=(r,)) for r in range(world)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            if errs:
                raise errs[0]
        solve_all()  # warm-up of the timed instance
        t0 = time.perf_counter()
        solve_all()
        t1 = time.perf_counter()
        recs = np.concatenate(parts)
        for bt in batches:
            bt.close()
        # the exchange step: every device ends up with all records (one rank per distinct device)
        uniq = sorted(set(devices))
        raw = np.ascontiguousarray(recs).view(np.uint8).reshape(total, -1)
        dv = (C.c_int * len(uniq))(*uniq)
        outp = (C.c_void_p * len(uniq))()
        L.check(core.lbfgsx_rccl_allgather_records(dv, len(uniq), raw.ctypes.data_as(C.c_void_p), total, raw.shape[1], outp))
        back = np.zeros_like(raw)
        L.check(core.lbfgsx_device_download(uniq[-1], outp[len(uniq) - 1], raw.size, back.ctypes.data_as(C.c_void_p)))
        for k, d in enumerate(uniq):
            core.lbfgsx_device_free(d, outp[k])
        t2 = time.perf_counter()
        if not np.array_equal(back, raw):
            raise SystemExit("bench.py --single-process: the gathered records differ from the solver's")
        its, fev = int(recs["niter"].sum()), int(recs["nfev"].sum())
        # the timed region is the batch itself; the exchange step is reported next to it (its time is ncclCommInitAll's:
        # the communicator lives for this one call)
        elapsed = t1 - t0
        hbm_ = (its * (4 * m + 2 + 12) + (fev - its) * 4) * n * 4.0
        model_gbs = hbm_ / elapsed / 1e9 / world
        out["cfg5_batched"] = {
            "metric": "batched L-BFGS problem-iterations/sec (cfg5: n=1e5, m=10, f32)", "value": its / elapsed,
            "unit": "problem-iterations/s", "n_gpus": world, "steps": steps, "ms_per_step": elapsed / max(steps, 1) * 1e3,
            "scaling": "weak", "dtype": "f32",
            "config": {"workload": "cfg5: %d independent extended-Rosenbrock problems per GPU, n=1e5, m=10, f32, "
                                   "LineSearchMoreThuente, %d iterations each; ONE process: one resident lock-step batch per device "
                                   "(lbfgsx_lockstep_*; contiguous problem-id blocks, one host thread per device) and the "
                                   "records all-gathered natively over RCCL" % (P, steps),
                       "problems_total": total, "fevals_total": fev, "failed": int((recs["status"] != 0).sum()),
                       "devices": devices, "solve_seconds": t1 - t0, "rccl_allgather_seconds": t2 - t1,
                       "rccl_ranks": len(uniq)},
            "roofline": {"bound": "hbm", "achieved": model_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": model_gbs / HBM_PEAK_GBS, "traffic": None,
                         "note": "end to end per GPU, HBM traffic model of the one-launch two-loop ((4m+14) n elements per "
                                 "iteration) / wall time of the batch; the RCCL gather of the records (communicator set-up "
                                 "included) is config.rccl_allgather_seconds"}}
    return out


def run_all(args, rank, world, local, comm_dev, dist, batched=True):
    """Everything one rank does for the chosen workload; the result object on rank 0 (None elsewhere)."""
    if args.workload == "cfg5-batched":
        out = run_batched(args, rank, world, local, comm_dev, dist, args.steps)
    else:
        out = run_north_star(args, rank, world, local, comm_dev, dist)
        if args.workload == "north-star" and args.recursion == "vector" and not args.no_batched and batched:
            # the mode that shards naturally: same ranks, same barriers, reported inside the one line
            b = run_batched(args, rank, world, local, comm_dev, dist, args.batched_steps)
            if rank == 0:
                out["cfg5_batched"] = {k: b[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step",
                                                          "scaling", "dtype", "config", "roofline")}
                out["cfg5_batched"]["kernel_ms_per_step"] = b["config"]["kernel_ms_per_step"]
        headline = int(args.n) == 100000000 and args.m == 10 and args.objective == "rosenbrock"
        if args.workload == "north-star" and args.recursion == "vector" and headline and not args.no_legs:
            # the other single-GPU configurations of BASELINE.json, each through the product path with its own roofline
            # (SURVEY 8(d)); every rank runs its own instance, like the headline
            leg = lbfgs_leg(args, rank, world, local, comm_dev, dist, objective="quadratic", n=1e7, m=10,
                            steps=max(args.steps, 20), warmup=12)
            if rank == 0:
                out["cfg2"] = leg
            leg = lbfgs_leg(args, rank, world, local, comm_dev, dist, objective="rosenbrock", n=1e8, m=20,
                            steps=min(args.steps, 10), warmup=22)
            if rank == 0:
                out["cfg3"] = leg
            leg = run_cfg4(args, rank, world, local, comm_dev, dist, iters=args.cfg4_iters)
            if rank == 0:
                out["cfg4_lbfgsb"] = leg
            if not args.no_cfg4_m20:
                # the same problem with the largest history a BASELINE configuration uses (cfg3's m = 20): the L-BFGS-B path
                # is generic in m, and this leg is what shows it; 60 iterations so that the steady window (31..59) runs on
                # a full history
                leg = run_cfg4(args, rank, world, local, comm_dev, dist, iters=max(args.cfg4_iters, 60), m=20)
                if rank == 0:
                    out["cfg4_m20"] = leg
        if rank == 0 and not args.no_cpu and world == 1:  # the CPU baseline is timed on rank 0 of the single-GPU run only
            try:
                out["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:  # the baseline is reported, never required
                out["cpu_baseline"] = {"error": repr(e)}
            if args.cpu_full != "off" and args.workload == "north-star":
                # the same reference at the metric's own size, measured: when it ran, IT is `cpu_baseline` and the scaled
                # small-n sample rides along as `cpu_baseline_sample`
                try:
                    full = cpu_baseline_full(args)
                except Exception as e:
                    full = {"error": repr(e)}
                if full and "value" in full:
                    out["cpu_baseline_sample"] = out["cpu_baseline"]
                    out["cpu_baseline"] = full
                else:
                    out["cpu_baseline_full"] = full
                ref = getattr(args, "_ref_run", None)
                if ref is not None:
                    try:
                        out["parity"] = north_star_parity(args, local, ref)
                    except Exception as e:
                        out["parity"] = {"error": repr(e)}
                    args._ref_run = None
                ref = getattr(args, "_ref_run_dd", None)
                if ref is not None:
                    try:
                        out["parity_dd"] = north_star_parity(args, local, ref)
                    except Exception as e:
                        out["parity_dd"] = {"error": repr(e)}
                    args._ref_run_dd = None
            try:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args)
            except Exception as e:
                out["cpu_baseline_all_cores"] = {"error": repr(e)}
    return out


# Keys that only the --verbose line carries: paragraph-long explanations and per-iteration lists.  The driver keeps the last
# 8 KB of the one JSON line; the default line is held under 7 KB and ends with `legs_digest`, so that every leg's figures are
# in whatever tail survives.
VERBOSE_ONLY = ("note", "per_iteration_ms", "traffic_source", "traffic_per", "achieved_is", "instance", "how")
# secondary figures of the headline's own objects that only the --verbose line carries
VERBOSE_ONLY_TOP = ("max_abs_dx_at", "x_inf_norm", "max_abs_dx_strided_sample", "stream_triad_GBs", "apply_Hv_GBs",
                    "hbm_model_bytes_per_launch", "fused_post_launches_timed", "warmup_run", "problems_per_gpu",
                    "algorithmic_GBs", "timed_iterations", "seconds_per_iteration", "measured_value")
FULL_PRECISION = ("fx", "value", "ms_per_step", "max_abs_dx", "fx_rel", "seconds")


def compact_line(o, key=None, depth=0):
    """The default line: explanations dropped below the top level (the headline's own stay), strings cut at 200 characters,
    floats to 7 significant digits except the ones compared digit by digit (FULL_PRECISION)."""
    if isinstance(o, dict):
        return {k: compact_line(v, k, depth + 1) for k, v in o.items()
                if not (depth >= 2 and k in VERBOSE_ONLY) and not (depth >= 1 and k in VERBOSE_ONLY_TOP)}
    if isinstance(o, list):
        return [compact_line(v, key, depth + 1) for v in o]
    if isinstance(o, float) and key not in FULL_PRECISION and not str(key).startswith("fx"):
        return float("%.7g" % o)
    if isinstance(o, str) and len(o) > 160 and depth > 1:
        return o[:157] + "..."
    return o


LEG_KEEP = {
    "": ("value", "unit", "steps", "warmup", "ms_per_step", "kernel_ms_per_step", "dtype", "from_x0", "config", "roofline"),
    "config": ("workload", "n", "m", "q", "n_free", "iterations", "fevals_total", "fx", "history_full",
               "problems_total", "failed", "host_syncs_per_iteration", "launches_per_iteration", "compact_passes_per_iteration",
               "one_launch_per_iteration", "launches_per_step", "host_waits_per_step", "setup_seconds"),
    "roofline": ("bound", "achieved", "peak", "unit", "frac", "frac_from_x0", "traffic", "traffic_frac", "traffic_GBs",
                 "model_bytes", "model_bytes_from_x0", "reference_statement_bytes", "avg_launch_ms", "hbm_model_GBs",
                 "algorithmic_GBs", "frac_of_stream_copy"),
    "from_x0": ("value", "first_iteration_ms", "host_syncs_per_iteration", "q"),
}


def compact_leg(leg):
    """A leg of the default line: the figures and what is needed to recompute them; everything else is --verbose."""
    if not isinstance(leg, dict):
        return leg
    out = {}
    for k in LEG_KEEP[""]:
        if k not in leg:
            continue
        v = leg[k]
        if k in LEG_KEEP and isinstance(v, dict):
            v = {kk: v[kk] for kk in LEG_KEEP[k] if kk in v}
            if isinstance(v.get("workload"), str) and len(v["workload"]) > 90:
                v["workload"] = v["workload"][:87] + "..."
        out[k] = v
    return out


LEGS = ("cfg2", "cfg3", "cfg4_lbfgsb", "cfg4_m20", "cfg5_batched")


def legs_digest(out):
    """{leg: value, ms_per_step, roofline fractions} -- the LAST key of the line."""
    dig = {}
    for name in ("north_star", "cfg2", "cfg3", "cfg4_lbfgsb", "cfg4_m20", "cfg5_batched"):
        leg = out if name == "north_star" else out.get(name)
        if not isinstance(leg, dict) or "value" not in leg:
            continue
        r = leg.get("roofline") or {}
        e = {"value": float("%.6g" % leg["value"]), "unit": leg.get("unit"), "ms_per_step": float("%.6g" % leg["ms_per_step"]),
             "frac": None if r.get("frac") is None else float("%.4g" % r["frac"])}
        if r.get("traffic_frac") is not None:
            e["traffic_frac"] = float("%.4g" % r["traffic_frac"])
        if leg.get("kernel_ms_per_step") is not None:  # cfg5: the kernels' share of ms_per_step
            e["kernel_ms_per_step"] = float("%.6g" % leg["kernel_ms_per_step"])
        if leg.get("value_median") is not None:  # cfg4 legs: value = 1 / MEAN iteration time of the window (rounds 1-3 quoted 1 / median)
            e["value_median"] = float("%.6g" % leg["value_median"])
        if isinstance(leg.get("from_x0"), dict):
            e["from_x0"] = float("%.6g" % leg["from_x0"]["value"])
            e["frac_from_x0"] = None if r.get("frac_from_x0") is None else float("%.4g" % r["frac_from_x0"])
        dig[name] = e
    return dig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs of this node (default: the launcher's WORLD_SIZE, else 1); N > 1 without a launcher starts N ranks")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--n", type=float, default=1e8)
    ap.add_argument("--m", type=int, default=10)
    ap.add_argument("--objective", default="rosenbrock", choices=["rosenbrock", "quadratic"])
    ap.add_argument("--cpu-n", type=float, default=2e7)
    ap.add_argument("--cpu-steps", type=int, default=6)
    ap.add_argument("--cpu-n-all", type=float, default=2e7, help="problem size of the all-cores CPU baseline")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-full", default="auto", choices=["auto", "on", "off"],
                    help="1-core reference at the metric's own n (m+2 warm-up + --cpu-full-steps timed iterations of one run, "
                         "un-extrapolated); auto: when the host has >= 48 GB available")
    ap.add_argument("--cpu-full-steps", type=int, default=3)
    ap.add_argument("--cpu-full-dd", action="store_true",
                    help="with the --cpu-full run: repeat it with the reference built on double-double sums and report the GPU "
                         "iterate against that one too (`parity_dd`; the expectation is max_abs_dx = 0)")
    ap.add_argument("--workload", default="north-star", choices=["north-star", "cfg5-batched", "sharded"],
                    help="north-star (default, the BASELINE.json metric: one problem per GPU; its line also carries the "
                         "cfg5 batch as `cfg5_batched`), the batched cfg5 shard per GPU alone, or sharded: ONE problem of "
                         "--n rows row-sharded over the ranks (strong scaling; opt-in Gram-space recursion, all-reduces "
                         "of <= 6m+7 doubles over RCCL)")
    ap.add_argument("--problems-per-gpu", type=int, default=1024)
    ap.add_argument("--batched-n", type=float, default=1e5, help="dimension of the batched leg's problems (cfg5: 1e5)")
    ap.add_argument("--batched-dtype", default="f32", choices=["f32", "f64"], help="scalar type of the batched leg (cfg5: f32)")
    ap.add_argument("--batched-steps", type=int, default=50, help="iterations per problem of the cfg5 leg of the default line")
    ap.add_argument("--no-batched", action="store_true", help="skip the cfg5 leg of the default line")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N from ONE process: one host thread + one context per GPU, the batch through "
                         "one resident lock-step batch per device and the record gather natively over RCCL")
    ap.add_argument("--verbose", action="store_true",
                    help="the full line (~15 KB: explanatory notes, per-iteration times); the default line drops them (< 7 KB)")
    ap.add_argument("--full-json", default=None, help="also write the full (verbose) object to this file")
    ap.add_argument("--no-legs", action="store_true", help="skip the cfg2 / cfg3 / cfg4 legs of the default line")
    ap.add_argument("--cfg4-n", type=float, default=1e7, help="problem size of the L-BFGS-B leg (cfg4)")
    ap.add_argument("--cfg4-iters", type=int, default=40, help="iterations from x0 of the L-BFGS-B leg (cfg4)")
    ap.add_argument("--no-cfg4-m20", action="store_true", help="skip the m = 20 repetition of the L-BFGS-B leg")
    ap.add_argument("--recursion", default="vector", choices=["vector", "gram", "gram-f32h"],
                    help="vector (default): the reference's two-loop recursion statement by statement, the bit-parity "
                         "path the BASELINE metric is quoted on; gram: opt-in Gram-space form (SURVEY 8(f)-3), equal to "
                         "the vector form only up to rounding; gram-f32h: the same with S, Y stored as float (SURVEY "
                         "8(f)-4) -- both reported under their own metric names")
    args = ap.parse_args()
    launched = "WORLD_SIZE" in os.environ
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.single_process and launched and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("bench.py: --single-process runs without a launcher (one process drives all GPUs)")
    if args.gpus > 1 and not launched and not args.single_process:
        spawn_ranks(args)  # does not return
    world_arg = args.gpus

    # Exactly ONE line on stdout: libraries that print there (gloo's connection notice, HIP runtime notes) are sent to
    # stderr for the whole run; the result line goes to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch  # noqa: F401  (first: its bundled HIP runtime must be the process-wide one)
    if args.single_process and world_arg > 1:
        rank, world = 0, 1
    else:
        rank, world, local, comm_dev, dist = init_dist(args)
    if args.single_process and world_arg > 1:
        out = run_single_process(args, world_arg)
    else:
        out = run_all(args, rank, world, local, comm_dev, dist)
    if rank == 0:
        sys.stdout.flush()
        if args.full_json:
            with open(args.full_json, "w") as f:
                json.dump(out, f, indent=1)
        if args.verbose:
            line = out
        else:
            line = dict(out)
            for name in LEGS:
                if name in line:
                    line[name] = compact_leg(line[name])
            if isinstance(line.get("cpu_baseline_sample"), dict):  # the scaled small-n sample beside the measured full-size one
                line["cpu_baseline_sample"] = {k: line["cpu_baseline_sample"].get(k) for k in ("value", "unit", "cores", "kind")}
            line = compact_line(line)
        line.pop("legs_digest", None)
        line["legs_digest"] = legs_digest(out)  # last key: survives a tail cut
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
