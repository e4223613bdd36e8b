#!/usr/bin/env python
"""Batched L-BFGS-B: independent box-constrained problems on ONE GPU, `nthreads` contexts resident at a time
(lbfgsx_batch_minimize with LBFGSX_ALGO_LBFGSB; include/lbfgsx_solver.h).  Problem-iterations per second by thread count.

    python scripts/bench_lbfgsb_batch.py [--n 100000] [--m 10] [--iters 30] [--count 64] [--threads 1,4,8,16]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=float, default=1e5)
    ap.add_argument("--m", type=int, default=10)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--count", type=int, default=64)
    ap.add_argument("--threads", default="1,4,8,16")
    ap.add_argument("--dtype", default="f64")
    args = ap.parse_args()
    import lbfgspp_amd as A
    from lbfgspp_amd import _lib as L
    from lbfgspp_amd import batched as B
    n = int(args.n)
    dt = np.float64 if args.dtype == "f64" else np.float32
    par = A.LBFGSBParam(m=args.m, epsilon=0.0, epsilon_rel=0.0, past=0, max_iterations=args.iters)
    B.solve_local(par, A.DiagQuadratic.objective, n, 0, 2, seed_base=1, algo=L.ALGO_LBFGSB, dtype=dt, nthreads=2)  # code objects
    out = {"workload": "independent L-BFGS-B problems (box quadratic [-1,1], kappa=10), n=%d, m=%d, %d iterations each, %d problems, %s"
                       % (n, args.m, args.iters, args.count, args.dtype), "runs": []}
    ref = None
    for t in [int(v) for v in args.threads.split(",")]:
        t0 = time.perf_counter()
        r = B.solve_local(par, A.DiagQuadratic.objective, n, 0, args.count, seed_base=100, algo=L.ALGO_LBFGSB, dtype=dt, nthreads=t)
        dt_s = time.perf_counter() - t0
        if ref is None:
            ref = r
        out["runs"].append({"nthreads": t, "seconds": dt_s, "problem_iterations_per_s": float(r["niter"].sum()) / dt_s,
                            "identical_to_first_run": bool(np.array_equal(r, ref)), "failed": int((r["status"] != 0).sum())})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
