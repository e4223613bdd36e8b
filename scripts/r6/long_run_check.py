#!/usr/bin/env python
"""a long L-BFGS-B run (200 iterations, not converged) next to the reference's: where the two trajectories part and who is lower"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import lbfgspp_amd as A
A.load()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rng = np.random.default_rng(9000 + seed)
kw = dict(m=int(rng.choice([3, 6, 10])), epsilon=float(rng.choice([0.0, 1e-8, 1e-4, 1e-2])),
          epsilon_rel=float(rng.choice([0.0, 1e-7, 1e-3])), past=int(rng.choice([0, 1, 2, 5])),
          delta=float(rng.choice([1e-14, 1e-8, 1e-4, 1e-1])), max_iterations=200)
n = int(rng.choice([10, 500, 3000]))
kappa = float(rng.choice([3.0, 30.0, 300.0]))
a, b = O.quad_problem(n, kappa, seed, O.F64)
lb, ub = -0.6 * np.ones(n), 0.8 * np.ones(n)
print("seed", seed, "n", n, "kappa", kappa, kw)
for env in ({}, {"LBFGSX_RHS_IDENTITY": "0"}, {"LBFGSX_RHS_IDENTITY": "0", "LBFGSX_GRAM_CARRY": "0", "LBFGSX_COMPLEMENT": "0"}):
    os.environ.update(env)
    ref = O.Oracle("ref")
    tr_ref = O.TraceBuf(n, cap=2048)
    x_ref, r_ref = ref.lbfgsb(O.F64, O.OBJ_QUAD, np.zeros(n), lb, ub, O.lbfgsb_params(**kw), a=a, b=b, trace=tr_ref)
    s = A.LBFGSBSolver(A.LBFGSBParam(**kw), dtype=np.float64)
    tr = A.TraceBuffer(n, cap=2048)
    x = np.zeros(n)
    s.minimize(A.DiagQuadratic(a, b), x, lb, ub, trace=tr)
    k = min(tr.count, tr_ref.count)
    d = np.abs(tr.xs[:k] - tr_ref.xs[:k]).max(axis=1)
    first = int(np.argmax(d > 1e-10)) if (d > 1e-10).any() else -1
    xs = np.clip(b / a, lb, ub)
    f = lambda v: float(0.5 * np.dot(a * v, v) - np.dot(b, v))
    print(env, "evals", tr.count, tr_ref.count, "niter", s.last.niter, r_ref.niter, "nfev", s.last.nfev, r_ref.nfev,
          "first eval with dx > 1e-10:", first, "dx there", d[first] if first >= 0 else 0, "max dx", d.max(),
          "\n   final fx ours %.15g ref %.15g  f(x*) %.15g   |x - x*| ours %.3g ref %.3g" % (s.last.fx, r_ref.fx, f(xs), np.abs(x - xs).max(), np.abs(x_ref - xs).max()))
    print("   dx at evals 10,20,40,60,80,120,160:", [float("%.2g" % d[min(i, k - 1)]) for i in (10, 20, 40, 60, 80, 120, 160)])
