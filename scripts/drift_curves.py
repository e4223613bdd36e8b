#!/usr/bin/env python
"""Drift of the GPU iterates from the CPU reference, evaluation by evaluation (test infrastructure: uses oracle/).

    python scripts/drift_curves.py cfg4   [--n 1e7] [--iters 8] [--devmin default,0]
    python scripts/drift_curves.py native [--n 200000] [--iters 40]

cfg4:   BASELINE.json config 4 (box quadratic [-1,1], m=10, f64, L-BFGS-B) against oracle/_ref (unmodified reference
        headers, extended-precision sums) at the benchmark's own size: max |x_gpu - x_ref| over the sampled coordinates
        of every objective evaluation and over all n final coordinates, for each hand-over point of the Cauchy search
        (LBFGSX_GCP_DEVICE_MIN; "default" = 4096 crossings on the host first, 0 = device from the first crossing,
        -1 = host only).  Reference loop: Cauchy.h:183-256.
native: the L-BFGS path (ext. Rosenbrock / More-Thuente and the diag quadratic / Nocedal-Wright) against the reference
        built with NATIVE accumulators (plain f64 sums in index order -- what Eigen's own reductions do up to their
        packet order), i.e. the drift the north_star tolerance "1e-10 against the Eigen reference" is about.
        Reference: BFGSMat.h:276-302, LBFGS.h:78-173.
One JSON object on stdout.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402


def per_eval(tr, tr_ref):
    k = min(tr.count, tr_ref.count)
    return [float(np.abs(tr.xs[i] - tr_ref.xs[i]).max()) for i in range(k)]


def cfg4(args):
    import lbfgspp_amd as A
    n, m, iters = int(args.n), args.m, args.iters
    stride = max(1, n // 20000)
    a, b = O.quad_problem(n, 10.0, 1, O.F64)
    lb, ub = -np.ones(n), np.ones(n)
    fam = "ref" if O.available("ref", "dd") else "port"
    orc = O.Oracle(fam, "dd")
    p = O.lbfgsb_params(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters)
    tr_ref = O.TraceBuf(n, cap=256, stride=stride)
    t0 = time.perf_counter()
    x_ref, r_ref = orc.lbfgsb(O.F64, O.OBJ_QUAD, np.zeros(n), lb, ub, p, a=a, b=b, trace=tr_ref)
    t_ref = time.perf_counter() - t0
    out = {"workload": "cfg4: box quadratic [-1,1], n=%d, m=%d, f64, LBFGSBSolver, %d iterations from x0=0" % (n, m, iters),
           "oracle": orc.description, "oracle_seconds": t_ref, "ref_niter": r_ref.niter, "ref_nfev": r_ref.nfev,
           "ref_fx": r_ref.fx, "sample_stride": stride, "runs": []}
    for dm in args.devmin.split(","):
        if dm == "default":
            os.environ.pop("LBFGSX_GCP_DEVICE_MIN", None)
        else:
            os.environ["LBFGSX_GCP_DEVICE_MIN"] = dm
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters))
        tr = A.TraceBuffer(n, cap=256, stride=stride)
        x = np.zeros(n)
        t0 = time.perf_counter()
        niter, fx = s.minimize(A.DiagQuadratic(a, b), x, lb, ub, trace=tr)
        t_gpu = time.perf_counter() - t0
        st = s.stats()
        out["runs"].append({
            "LBFGSX_GCP_DEVICE_MIN": dm, "niter": niter, "nfev": s.last.nfev, "fx": fx,
            "same_counts": bool(niter == r_ref.niter and s.last.nfev == r_ref.nfev),
            "fx_rel_diff": abs(fx - r_ref.fx) / abs(r_ref.fx),
            "max_dx_per_evaluation": per_eval(tr, tr_ref),
            "max_dx_final_all_coordinates": float(np.abs(x - x_ref).max()),
            "same_active_set": bool(np.array_equal(np.abs(x) == 1.0, np.abs(x_ref) == 1.0)),
            "gcp_crossings": st["gcp_crossings"], "gcp_dev_crossings": st["gcp_dev_crossings"],
            "submin_sweeps": st["submin_sweeps"], "seconds_incl_transfers": t_gpu})
        s.close()
    return out


def native(args):
    import lbfgspp_amd as A
    n, iters = int(args.n), args.iters
    assert O.available("ref", "native"), "oracle/_ref/libref_native.so missing (make -C oracle ref)"
    nat, dd = O.Oracle("ref", "native"), O.Oracle("ref", "dd")
    out = {"reference": nat.description, "n": n, "curves": []}
    cases = [("ext. Rosenbrock, m=10, LineSearchMoreThuente (north-star objective)", O.OBJ_ROSEN, O.LS_MT, A.LS_MORE_THUENTE, 10),
             ("diag quadratic kappa=10, m=10, LineSearchNocedalWright (cfg2 objective)", O.OBJ_QUAD, O.LS_NW, A.LS_NOCEDAL_WRIGHT, 10)]
    for name, obj, ls, als, m in cases:
        p = O.lbfgs_params(m=m, epsilon=0, epsilon_rel=0, max_iterations=iters)
        if obj == O.OBJ_ROSEN:
            x0, a, b, f = O.rosen_x0(n), None, None, A.ExtendedRosenbrock()
        else:
            a, b = O.quad_problem(n, 10.0, 1, O.F64)
            x0, f = np.zeros(n), A.DiagQuadratic(a, b)
        tn, td, tg = O.TraceBuf(n, cap=512), O.TraceBuf(n, cap=512), A.TraceBuffer(n, cap=512)
        xn, rn = nat.lbfgs(O.F64, ls, obj, x0, p, a=a, b=b, trace=tn)
        xd, rd = dd.lbfgs(O.F64, ls, obj, x0, p, a=a, b=b, trace=td)
        s = A.LBFGSSolver(A.LBFGSParam(m=m, epsilon=0, epsilon_rel=0, max_iterations=iters), linesearch=als)
        x = x0.copy()
        niter, fx = s.minimize(f, x, trace=tg)
        scale = float(np.abs(xn).max())
        out["curves"].append({
            "case": name, "iterations": niter, "evaluations": tg.count, "x_scale": scale,
            "counts_equal_native": bool(niter == rn.niter and s.last.nfev == rn.nfev),
            "gpu_vs_native_reference": per_eval(tg, tn),
            "gpu_vs_extended_reference": per_eval(tg, td),
            "extended_vs_native_reference": per_eval(td, tn),
            "final_gpu_vs_native": float(np.abs(x - xn).max())})
        s.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["cfg4", "native"])
    ap.add_argument("--n", type=float, default=None)
    ap.add_argument("--m", type=int, default=10)
    ap.add_argument("--iters", type=int, default=None)
    ap.add_argument("--devmin", default="default,0")
    args = ap.parse_args()
    if args.mode == "cfg4":
        args.n = args.n or 1e7
        args.iters = args.iters or 8
        res = cfg4(args)
    else:
        args.n = args.n or 200000
        args.iters = args.iters or 40
        res = native(args)
    print(json.dumps(res))
