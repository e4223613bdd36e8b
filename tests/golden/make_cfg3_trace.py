#!/usr/bin/env python
"""Oracle side of the at-size cfg3 parity test, computed once and committed (test infrastructure).

BASELINE.json cfg3 -- extended Rosenbrock, n = 1e8, m = 20, f64, LineSearchMoreThuente, the counter-hash start point bench.py
times (oracle_lib.rosen_x0(n, seed 7) = lbfgsx_gen_rosen_x0(ctx, 7)) -- through the UNMODIFIED reference headers built on the
parity contract's extended sums (oracle/_ref/libref_dd.so, /root/reference/include/LBFGS.h:78-173) needs ~45 GB of host
memory and minutes of one core: more than a GPU test may spend.  This script runs it here for ITERS iterations (no GPU needed)
and stores what the test compares: the objective value and a strided sample of x at every evaluation, a finer sample of the
final x, the counts and the final gradient norm.  Keyed by the oracle's build key (oracle/_ref/build_key.txt), like
cfg4_1e7_trace.npz: tests/test_full_size_gpu.py uses it only while the key matches the library it would otherwise have called.

    python tests/golden/make_cfg3_trace.py [--iters 12]
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

N, M, STRIDE, FINAL_STRIDE = 100_000_000, 20, 40_000, 2_500


def build_key():
    p = os.path.join(ROOT, "oracle", "_ref", "build_key.txt")
    return open(p).read().strip() if os.path.exists(p) else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=12)
    args = ap.parse_args()
    key = build_key()
    if key is None or not O.available("ref", "dd"):
        raise SystemExit("oracle/_ref is not built (make -C oracle ref)")
    orc = O.Oracle("ref", "dd")
    p = O.lbfgs_params(m=M, epsilon=0, epsilon_rel=0, past=0, max_iterations=args.iters)
    tr = O.TraceBuf(N, cap=256, stride=STRIDE)
    x0 = O.rosen_x0(N)
    t0 = time.perf_counter()
    x, r = orc.lbfgs(O.F64, O.LS_MT, O.OBJ_ROSEN, x0, p, trace=tr)
    dt = time.perf_counter() - t0
    k = tr.count
    out = os.path.join(HERE, "cfg3_1e8_m20_trace.npz")
    np.savez(out, key=np.array(key), n=N, m=M, iters=args.iters, stride=STRIDE, final_stride=FINAL_STRIDE, niter=r.niter,
             nfev=r.nfev, fx=r.fx, gnorm=r.gnorm, fx_per_eval=tr.fx[:k].copy(), xs=tr.xs[:k].copy(),
             x_final=x[::FINAL_STRIDE].copy(), oracle=np.array(orc.description), seconds=dt)
    print("wrote %s: %d iterations, %d evaluations, fx = %.17g, %.0f s" % (out, r.niter, r.nfev, r.fx, dt))


if __name__ == "__main__":
    main()
