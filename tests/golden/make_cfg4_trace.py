#!/usr/bin/env python
"""Oracle side of the at-size cfg4 parity test, computed once and committed (test infrastructure).

BASELINE.json cfg4 -- L-BFGS-B on the box quadratic, n = 1e7, m = 10, lb = -1, ub = 1, x0 = 0, f64 -- through the UNMODIFIED
reference headers (oracle/_ref/libref_dd.so, /root/reference/include/LBFGSB.h:116-262) takes about 8 s per iteration on one
host core, so the GPU test can only afford a handful of iterations when it runs the oracle itself.  This script runs it for
ITERS iterations here (no GPU needed) and stores what the test compares: the objective value and a strided sample of x at
every evaluation, a finer sample of the final x, the counts and the size of the active set.  The file is keyed by the
oracle's build key (oracle/_ref/build_key.txt: a hash of the reference headers, the stand-in Eigen, the driver and the
compiler flags, written by oracle/Makefile): tests/test_gcp_device_gpu.py uses it only while the key matches the library
it would otherwise have called.

    python tests/golden/make_cfg4_trace.py [--iters 40] [--m 10]

Round 5: the default is the benchmark's own 40 iterations (62 evaluations, ~5 minutes of one core), so that the always-on
GPU test covers every iteration bench.py times; `--m 20 --iters 30` writes cfg4_1e7_m20_trace.npz for the m = 20 leg.
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

N, STRIDE, FINAL_STRIDE = 10_000_000, 4000, 500


def build_key():
    p = os.path.join(ROOT, "oracle", "_ref", "build_key.txt")
    return open(p).read().strip() if os.path.exists(p) else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--m", type=int, default=10)
    args = ap.parse_args()
    M = args.m
    key = build_key()
    if key is None or not O.available("ref", "dd"):
        raise SystemExit("oracle/_ref is not built (make -C oracle ref)")
    orc = O.Oracle("ref", "dd")
    a, b = O.quad_problem(N, 10.0, 1, O.F64)
    lb, ub = -np.ones(N), np.ones(N)
    p = O.lbfgsb_params(m=M, epsilon=0, epsilon_rel=0, past=0, max_iterations=args.iters)
    tr = O.TraceBuf(N, cap=256, stride=STRIDE)
    t0 = time.perf_counter()
    x, r = orc.lbfgsb(O.F64, O.OBJ_QUAD, np.zeros(N), lb, ub, p, a=a, b=b, trace=tr)
    dt = time.perf_counter() - t0
    k = tr.count
    out = os.path.join(HERE, "cfg4_1e7_trace.npz" if M == 10 else "cfg4_1e7_m%d_trace.npz" % M)
    np.savez(out, key=np.array(key), n=N, m=M, iters=args.iters, stride=STRIDE, final_stride=FINAL_STRIDE, niter=r.niter,
             nfev=r.nfev, fx=r.fx, fx_per_eval=tr.fx[:k].copy(), xs=tr.xs[:k].copy(), x_final=x[::FINAL_STRIDE].copy(),
             n_active=int((np.abs(x) == 1.0).sum()), active_sample=(np.abs(x[::FINAL_STRIDE]) == 1.0),
             oracle=np.array(orc.description), seconds=dt)
    print("%s: %d iterations, %d evaluations, %.0f s of one core, %.0f KB" % (out, r.niter, k, dt, os.path.getsize(out) / 1e3))


if __name__ == "__main__":
    main()
