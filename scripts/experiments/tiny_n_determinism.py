"""Experiment: are tiny problems (n = 2 .. 24) solved identically run after run, and identically with the persistent
launch and the step launches?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lbfgspp_amd as A
from lbfgspp_amd import _lib as L

def run(n, ls, x0s, persist):
    os.environ["LBFGSX_PERSIST"] = persist
    par = A.LBFGSParam(max_iterations=200, max_linesearch=256)
    out = []
    s = A.LBFGSSolver(par, linesearch=ls)
    for x0 in x0s:
        x = x0.copy()
        try:
            niter, fx = s.minimize(A.ExtendedRosenbrock(), x)
            out.append((niter, s.last.nfev, fx, x.tobytes()))
        except Exception as e:
            out.append(("exc", s.last.nfev, str(e)[:30], b""))
    s.close()
    return out

rng = np.random.default_rng(1)
for n in (2, 4, 10, 24):
    x0s = [rng.uniform(-1, 1, n) for _ in range(300)]
    for ls, name in ((L.LS_BACKTRACKING, "backtracking"), (L.LS_NOCEDAL_WRIGHT, "nocedal-wright")):
        a = run(n, ls, x0s, "1")
        b = run(n, ls, x0s, "1")
        c = run(n, ls, x0s, "0")
        d = run(n, ls, x0s, "0")
        print("n=%2d %-15s persist vs persist: %3d differ | step vs step: %3d | persist vs step: %3d | calls %d / %d"
              % (n, name, sum(u != v for u, v in zip(a, b)), sum(u != v for u, v in zip(c, d)), sum(u != v for u, v in zip(a, c)),
                 sum(u[1] for u in a), sum(u[1] for u in c)))
