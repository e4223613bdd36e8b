import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import lbfgspp_amd as A
import oracle_lib as O
which = sys.argv[1]
n = int(sys.argv[2])
if which == "quad":
    a, b = O.quad_problem(n)
    f = A.DiagQuadratic(a, b); ls = A.LS_NOCEDAL_WRIGHT; x = np.zeros(n)
elif which == "quadmt":
    a, b = O.quad_problem(n)
    f = A.DiagQuadratic(a, b); ls = A.LS_MORE_THUENTE; x = np.zeros(n)
else:
    f = A.ExtendedRosenbrock(); ls = A.LS_MORE_THUENTE; x = O.rosen_x0(n)
s = A.LBFGSSolver(A.LBFGSParam(m=10, epsilon=0, epsilon_rel=0, max_iterations=12), linesearch=ls)
print("start", which, n, flush=True)
niter, fx = s.minimize(f, x)
print("done", niter, fx, flush=True)
