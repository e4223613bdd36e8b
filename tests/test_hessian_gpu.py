"""-m gpu: LBFGSSolver::final_approx_hessian() / final_approx_inverse_hessian() (reference LBFGS.h:192-197,
BFGSMat.h:150-271) against the committed fixture generated from oracle/_ref and against the live oracle."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def A():
    import lbfgspp_amd as A
    A.load()
    return A


def _cases():
    with open(os.path.join(HERE, "golden", "hessian_golden.json")) as f:
        return json.load(f)["cases"]


def _solve(A, c):
    n = c["n"]
    s = A.LBFGSSolver(A.LBFGSParam(m=c["m"], max_iterations=c["max_iterations"], epsilon=1e-6), linesearch=c["ls"])
    if c["obj"] == O.OBJ_ROSEN:
        x = O.rosen_x0(n, 7, O.F64) if c["hash_x0"] else np.zeros(n)
        f = A.ExtendedRosenbrock()
    else:
        x = np.zeros(n)
        f = A.DiagQuadratic(*O.quad_problem(n, 10.0, 1, O.F64))
    niter, _ = s.minimize(f, x)
    B, H = s.final_approx_hessians(n)
    return niter, B, H


@pytest.mark.parametrize("c", _cases(), ids=lambda c: c["name"])
def test_dense_hessians_match_reference_fixture(A, c):
    n = c["n"]
    niter, B, H = _solve(A, c)
    assert niter == c["niter"]
    Bg = np.array([float.fromhex(v) for v in c["B"]]).reshape(n, n, order="F")
    Hg = np.array([float.fromhex(v) for v in c["H"]]).reshape(n, n, order="F")
    # tolerance: north_star's 1e-10 relative to the matrix scale (the dense getters go through a small LU solve)
    assert np.abs(B - Bg).max() <= 1e-10 * np.abs(Bg).max()
    assert np.abs(H - Hg).max() <= 1e-10 * np.abs(Hg).max()
    assert np.abs(B - B.T).max() <= 1e-12 * np.abs(B).max()
    assert np.abs(B @ H - np.eye(n)).max() < 1e-8


def test_dense_inverse_hessian_is_the_two_loop_operator(A, oracle):
    """H from the getter applied to v == apply_Hv(v) on the same history (BFGSMat.h:276-302)."""
    c = _cases()[1]
    n = c["n"]
    s = A.LBFGSSolver(A.LBFGSParam(m=c["m"], max_iterations=c["max_iterations"], epsilon=1e-6), linesearch=c["ls"])
    x = O.rosen_x0(n, 7, O.F64)
    s.minimize(A.ExtendedRosenbrock(), x)
    B, H = s.final_approx_hessians(n)
    live = oracle.lbfgs_hessians(O.F64, c["ls"], c["obj"], O.rosen_x0(n, 7, O.F64),
                                 O.lbfgs_params(m=c["m"], max_iterations=c["max_iterations"], epsilon=1e-6))
    if live is not None:
        assert np.abs(H - live[3]).max() <= 1e-10 * np.abs(live[3]).max()
        assert np.abs(B - live[2]).max() <= 1e-10 * np.abs(live[2]).max()
    v = np.linspace(-1.0, 2.0, n)
    # secant equation on the newest pair is implied by B H = I; check H v through the definition instead
    assert np.allclose(B @ (H @ v), v, rtol=0, atol=1e-8)
