#!/usr/bin/env python
"""cfg4 of BASELINE.json: L-BFGS-B on the box-constrained diag quadratic, lb=-1, ub=1 (run on the GPU box)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=float, default=1e7)
    ap.add_argument("--m", type=int, default=10)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--cpu-n", type=float, default=0)
    ap.add_argument("--no-warmup", action="store_true",
                    help="skip the untimed small solve that loads the code objects (first-launch cost of ~60 kernels)")
    args = ap.parse_args()
    import torch  # noqa: F401
    import lbfgspp_amd as A
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    n = int(args.n)
    if not args.no_warmup:
        # untimed warm-up on a small instance of the same problem: HIP loads a kernel's code object at its first launch
        w = A.LBFGSBSolver(A.LBFGSBParam(m=args.m, epsilon=0, epsilon_rel=0, past=0, max_iterations=12))
        wn = 1 << 18
        wctx = w.prepare(wn)
        L.check(core.lbfgsx_gen_diag_quad(wctx, 10.0, 1))
        L.check(core.lbfgsx_fill(wctx, L.VEC_X, 0.0))
        L.check(core.lbfgsx_fill(wctx, L.VEC_LB, -1.0))
        L.check(core.lbfgsx_fill(wctx, L.VEC_UB, 1.0))
        w.minimize_resident(A.DiagQuadratic(), wn)
        w.close()
    s = A.LBFGSBSolver(A.LBFGSBParam(m=args.m, epsilon=0, epsilon_rel=0, past=0, max_iterations=args.iters))
    ctx = s.prepare(n)
    L.check(core.lbfgsx_gen_diag_quad(ctx, 10.0, 1))
    L.check(core.lbfgsx_fill(ctx, L.VEC_X, 0.0))
    L.check(core.lbfgsx_fill(ctx, L.VEC_LB, -1.0))
    L.check(core.lbfgsx_fill(ctx, L.VEC_UB, 1.0))
    L.check(core.lbfgsx_sync(ctx))
    import ctypes as C
    stamps, csnaps = [], []
    cnt = (C.c_int64 * 8)()

    def hook(k):
        stamps.append(time.perf_counter())
        core.lbfgsx_counters_ex(C.byref(cnt), 0)
        csnaps.append(tuple(cnt[i] for i in range(6)))
    s.set_iteration_hook(hook)
    core.lbfgsx_counters_ex(None, 1)
    t0 = time.perf_counter()
    niter, fx = s.minimize_resident(A.DiagQuadratic(), n)
    t1 = time.perf_counter()
    per = np.diff(np.array([t0] + stamps))
    out = dict(n=n, m=args.m, warmup=not args.no_warmup, niter=niter, nfev=s.last.nfev, fx=fx, total_s=t1 - t0, it_per_s=niter / (t1 - t0),
               steady_it_per_s=float(1.0 / np.median(per[len(per) // 2:])) if len(per) > 4 else None,
               per_iter_ms=[round(1e3 * v, 2) for v in per], stats=s.stats())
    if len(csnaps) >= 4:
        # the byte model of the path as built (lbfgsx_counters_ex) over the steady window = the second half of the iterations,
        # and what it gives against the mean iteration time of the same window (bench.py's cfg4 leg does the same)
        w0 = len(csnaps) // 2
        a_c, b_c = csnaps[w0 - 1], csnaps[-1]
        nwin = len(csnaps) - w0
        win_s = float(np.mean(per[w0:w0 + nwin]))
        out["model"] = dict(bytes_per_iteration=(b_c[3] - a_c[3]) / nwin, compact_passes_per_iteration=(b_c[4] - a_c[4]) / nwin,
                            n_free=(b_c[5] - a_c[5]) / max(1, b_c[4] - a_c[4]), launches_per_iteration=(b_c[0] - a_c[0]) / nwin,
                            host_syncs_per_iteration=(b_c[1] - a_c[1]) / nwin, window_ms_per_iteration=win_s * 1e3,
                            model_GBs=(b_c[3] - a_c[3]) / nwin / win_s / 1e9, frac=(b_c[3] - a_c[3]) / nwin / win_s / 8e12,
                            bytes_per_iteration_from_x0=csnaps[-1][3] / len(csnaps))
    pc = (C.c_int64 * 2)()
    core.lbfgsx_poll_counts(ctx, C.byref(pc))
    out["polled_waits"], out["poll_timeouts"] = int(pc[0]), int(pc[1])
    # algorithmic bytes of an L-BFGS-B iteration with q BOXCQP sweeps and one objective evaluation,
    # [(4m + 19) + (q + 1)(4m + 1)] n elements (history full); the roofline view of the steady state against 8 TB/s
    st = out["stats"]
    q = st["submin_sweeps"] / max(1, st["submin_calls"])
    bytes_it = ((4 * args.m + 19) + (q + 1.0) * (4 * args.m + 1)) * n * 8
    out["q_sweeps_per_iteration"] = q
    out["gcp_crossings_per_iteration"] = st["gcp_crossings"] / max(1, niter)
    out["algorithmic_bytes_per_iteration"] = bytes_it
    if out["steady_it_per_s"]:
        ach = bytes_it * out["steady_it_per_s"] / 1e9
        out["roofline"] = dict(bound="hbm", achieved=ach, peak=8000.0, unit="GB/s", frac=ach / 8000.0,
                               note="steady state (median of the second half of the iterations), algorithmic bytes / time")
    if args.cpu_n:
        import oracle_lib as O
        orc = O.Oracle("ref", "native")
        cn = int(args.cpu_n)
        a, b = O.quad_problem(cn)
        p = O.lbfgsb_params(m=args.m, epsilon=0, epsilon_rel=0, past=0, max_iterations=args.iters)
        t = time.perf_counter()
        _, r = orc.lbfgsb(O.F64, O.OBJ_QUAD, np.zeros(cn), -np.ones(cn), np.ones(cn), p, a=a, b=b)
        dt = time.perf_counter() - t
        out["cpu_reference"] = dict(n=cn, niter=r.niter, nfev=r.nfev, seconds=dt, it_per_s=r.niter / dt)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
