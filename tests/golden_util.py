"""Shared helpers to replay tests/golden/*.json cases against any implementation."""
import json
import os

import numpy as np

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name="lbfgs_golden.json"):
    with open(os.path.join(HERE, "golden", name)) as f:
        return json.load(f)


def unhex(lst):
    return np.array([float.fromhex(v) for v in lst], dtype=np.float64)


def case_inputs(c):
    dtype, n = c["dtype"], c["n"]
    if c["obj"] == O.OBJ_ROSEN:
        x0 = O.rosen_x0(n, c["seed"], dtype) if c["hash_x0"] else np.zeros(n, O.NPDT[dtype])
        a = b = None
    else:
        x0 = np.zeros(n, O.NPDT[dtype])
        a, b = O.quad_problem(n, c["kappa"], 1, dtype)
    return x0, a, b
