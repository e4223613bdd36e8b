import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
import lbfgspp_amd as A
from lbfgspp_amd import batched as B
par = A.LBFGSParam(m=10, epsilon=0.0, epsilon_rel=0.0, max_iterations=50)
B.solve_local_lockstep(par, 100000, 0, 8, dtype=np.float32)
t=time.perf_counter(); r=B.solve_local_lockstep(par, 100000, 0, 1024, dtype=np.float32); print("1024:", time.perf_counter()-t)
t=time.perf_counter(); r=B.solve_local_lockstep(par, 100000, 0, 1024, dtype=np.float32); print("1024 again:", time.perf_counter()-t)
