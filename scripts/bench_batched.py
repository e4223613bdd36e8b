#!/usr/bin/env python
"""cfg5 of BASELINE.json: independent extended-Rosenbrock problems, n=1e5, m=10, f32, fixed iteration budget.
Per GPU: --count problems (1024 in cfg5).  Reports problem-iterations/s and the implied HBM rate."""
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
    ap.add_argument("--count", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--threads", type=str, default="1,4,16,32")
    args = ap.parse_args()
    import torch  # noqa: F401
    import lbfgspp_amd as A
    from lbfgspp_amd import batched as B
    n = int(args.n)
    par = A.LBFGSParam(m=args.m, epsilon=0.0, epsilon_rel=0.0, max_iterations=args.iters)
    out = dict(n=n, m=args.m, count=args.count, iters=args.iters, runs=[])
    B.solve_local(par, A.ExtendedRosenbrock.objective, n, 0, 4, dtype=np.float32, nthreads=4)  # warm-up
    for t in [int(v) for v in args.threads.split(",")]:
        t0 = time.perf_counter()
        recs = B.solve_local(par, A.ExtendedRosenbrock.objective, n, 0, args.count, seed_base=1000, dtype=np.float32,
                             nthreads=t)
        dt = time.perf_counter() - t0
        its = int(recs["niter"].sum())
        fev = int(recs["nfev"].sum())
        # algorithmic bytes per iteration (8m+12) n 4 B plus 4 n 4 B per extra line-search trial
        bytes_ = (its * (8 * args.m + 12) + (fev - its) * 4) * n * 4.0
        out["runs"].append(dict(threads=t, seconds=dt, problem_iterations_per_s=its / dt, fevals=fev, iterations=its,
                                failed=int((recs["status"] != 0).sum()), algorithmic_GBs=bytes_ / dt / 1e9))
    t0 = time.perf_counter()
    recs = B.solve_local_lockstep(par, n, 0, args.count, seed_base=1000, dtype=np.float32)
    dt = time.perf_counter() - t0
    its, fev = int(recs["niter"].sum()), int(recs["nfev"].sum())
    bytes_ = (its * (8 * args.m + 12) + (fev - its) * 4) * n * 4.0
    out["lockstep"] = dict(seconds=dt, problem_iterations_per_s=its / dt, fevals=fev, iterations=its,
                           failed=int((recs["status"] != 0).sum()), algorithmic_GBs=bytes_ / dt / 1e9)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
