"""CPU suite, part 1: the oracle itself.

* the restatement (oracle/lbfgs_oracle.cpp) must reproduce every committed golden fixture bit for bit;
  the fixtures were generated from the reference itself (unmodified headers + eigen_shim), see
  tests/golden/make_golden.py;
* where oracle/_ref is present (build container) the restatement and the reference-derived oracle must agree
  bit for bit on fresh seeded inputs, and the double-double build must agree with the __float128 build.
"""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O

GOLD = G.load()


def _port():
    if not O.available("port", "dd"):
        pytest.skip("oracle port not built (make -C oracle port)")
    return O.Oracle("port", "dd")


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_restatement_reproduces_golden(case):
    orc = _port()
    x0, a, b = G.case_inputs(case)
    p = O.Params(**case["params"])
    tr = O.TraceBuf(case["n"], cap=1024, stride=case["stride"])
    x, r = orc.lbfgs(case["dtype"], case["ls"], case["obj"], x0, p, a=a, b=b, trace=tr)
    assert (r.niter, r.nfev, r.status) == (case["niter"], case["nfev"], case["status"])
    assert r.fx == float.fromhex(case["fx"]) and r.gnorm == float.fromhex(case["gnorm"])
    k = tr.count
    assert np.array_equal(tr.fx[:k], G.unhex(case["trace_fx"]))
    assert np.array_equal(tr.xs[:k].ravel(), G.unhex(case["trace_xs"]))
    assert np.array_equal(np.asarray(x[::case["stride"]], np.float64), G.unhex(case["x_sample"]))


def test_known_answers_from_reference_docs():
    """iteration / evaluation counts of the README problem for the four line searches."""
    by = {c["name"]: c for c in GOLD["cases"]}
    assert (by["readme_rosen10_f64_nw"]["niter"], by["readme_rosen10_f64_nw"]["nfev"]) == (22, 36)
    assert (by["readme_rosen10_f64_mt"]["niter"], by["readme_rosen10_f64_mt"]["nfev"]) == (21, 28)
    assert (by["readme_rosen10_f64_bt"]["niter"], by["readme_rosen10_f64_bt"]["nfev"]) == (22, 31)
    assert (by["readme_rosen10_f64_br"]["niter"], by["readme_rosen10_f64_br"]["nfev"]) == (22, 31)
    # README.md:89-95 "23 iterations" is reproduced with epsilon_rel = 0 (the README predates epsilon_rel)
    assert by["readme_rosen10_f64_nw_epsrel0"]["niter"] == 23


@pytest.mark.skipif(not O.available("ref", "dd"), reason="oracle/_ref only exists where /root/reference does")
@pytest.mark.parametrize("dtype", [O.F64, O.F32])
@pytest.mark.parametrize("ls", [O.LS_NW, O.LS_MT, O.LS_BT, O.LS_BR])
def test_restatement_equals_reference_build(dtype, ls):
    ref, port = O.Oracle("ref", "dd"), _port()
    for obj, n in ((O.OBJ_ROSEN, 1500), (O.OBJ_QUAD, 2001)):
        x0 = O.rosen_x0(n, 11, dtype) if obj == O.OBJ_ROSEN else np.zeros(n, O.NPDT[dtype])
        a, b = O.quad_problem(n, 50.0, 3, dtype) if obj == O.OBJ_QUAD else (None, None)
        p = O.lbfgs_params(m=7, epsilon=0, epsilon_rel=0, max_iterations=40)
        t1, t2 = O.TraceBuf(n, cap=600), O.TraceBuf(n, cap=600)
        x1, r1 = ref.lbfgs(dtype, ls, obj, x0, p, a=a, b=b, trace=t1)
        x2, r2 = port.lbfgs(dtype, ls, obj, x0, p, a=a, b=b, trace=t2)
        assert (r1.niter, r1.nfev, r1.status, r1.msg) == (r2.niter, r2.nfev, r2.status, r2.msg)
        assert np.array_equal(x1, x2) and np.array_equal(t1.xs[:t1.count], t2.xs[:t2.count])


@pytest.mark.skipif(not (O.available("ref", "dd") and O.available("ref", "quad")), reason="needs oracle/_ref")
def test_double_double_equals_float128_accumulation():
    """The parity oracle's double-double reductions give the same trajectories as __float128 reductions:
    both are correctly rounded sums, hence independent of the summation order."""
    dd, quad = O.Oracle("ref", "dd"), O.Oracle("ref", "quad")
    n = 6000
    x0 = O.rosen_x0(n)
    p = O.lbfgs_params(m=8, epsilon=0, epsilon_rel=0, max_iterations=40)
    x1, r1 = dd.lbfgs(O.F64, O.LS_MT, O.OBJ_ROSEN, x0, p)
    x2, r2 = quad.lbfgs(O.F64, O.LS_MT, O.OBJ_ROSEN, x0, p)
    assert (r1.niter, r1.nfev) == (r2.niter, r2.nfev) and np.array_equal(x1, x2)
    a, b = O.quad_problem(n)
    pb = O.lbfgsb_params(m=8, epsilon=0, epsilon_rel=0, max_iterations=15, past=0)
    x1, r1 = dd.lbfgsb(O.F64, O.OBJ_QUAD, np.zeros(n), -np.ones(n), np.ones(n), pb, a=a, b=b)
    x2, r2 = quad.lbfgsb(O.F64, O.OBJ_QUAD, np.zeros(n), -np.ones(n), np.ones(n), pb, a=a, b=b)
    assert (r1.niter, r1.nfev) == (r2.niter, r2.nfev) and np.array_equal(x1, x2)


@pytest.mark.skipif(not O.available("ref", "dd"), reason="needs oracle/_ref")
def test_apply_Hv_restatement_equals_reference():
    ref, port = O.Oracle("ref", "dd"), _port()
    rng = np.random.default_rng(5)
    for dtype in (O.F64, O.F32):
        for n, m, k in ((257, 6, 0), (257, 6, 4), (1001, 5, 12)):
            S = rng.standard_normal((max(k, 1), n))[:k]
            Y = (S * (1 + rng.random((k, n)))) if k else S
            v = rng.standard_normal(n)
            assert np.array_equal(ref.apply_Hv(dtype, m, S.reshape(k, n), Y.reshape(k, n), v, -1.0),
                                  port.apply_Hv(dtype, m, S.reshape(k, n), Y.reshape(k, n), v, -1.0))


# ---------------------------------------------------------------- L-BFGS-B part of the restatement
GOLDB = G.load("lbfgsb_golden.json")


def _gold_instance(seed, n, npairs, mode):
    from golden.make_golden import lbfgsb_instance
    return lbfgsb_instance(seed, n, npairs, mode)


@pytest.mark.parametrize("case", GOLDB["trajectories"], ids=[c["name"] for c in GOLDB["trajectories"]])
def test_lbfgsb_restatement_reproduces_golden_trajectory(case):
    orc = _port()
    n = case["n"]
    a, b = O.quad_problem(n)
    p = O.lbfgsb_params(m=case["m"], epsilon=0.0, epsilon_rel=0.0, past=0, max_iterations=case["max_iterations"])
    tr = O.TraceBuf(n, cap=1024, stride=case["stride"])
    x, r = orc.lbfgsb(O.F64, O.OBJ_QUAD, np.zeros(n), -np.ones(n), np.ones(n), p, a=a, b=b, trace=tr)
    assert (r.niter, r.nfev, r.status) == (case["niter"], case["nfev"], 0)
    k = tr.count
    assert np.array_equal(tr.xs[:k].ravel(), G.unhex(case["trace_xs"]))
    assert np.array_equal(x[::case["stride"]], G.unhex(case["x_sample"]))
    assert r.fx == float.fromhex(case["fx"]) and r.gnorm == float.fromhex(case["gnorm"])


@pytest.mark.parametrize("inst", GOLDB["instances"], ids=["seed%d" % i["seed"] for i in GOLDB["instances"]])
def test_cauchy_subspace_restatement_reproduces_golden(inst):
    orc = _port()
    S, Y, x0, g, lb, ub = _gold_instance(inst["seed"], inst["n"], inst["npairs"], inst["mode"])
    res = orc.cauchy_subspace(O.F64, inst["m"], S, Y, x0, g, lb, ub, max_submin=10)
    assert np.array_equal(res["xcp"], G.unhex(inst["xcp"])) and np.array_equal(res["drt"], G.unhex(inst["drt"]))
    assert np.array_equal(res["vecc"], G.unhex(inst["vecc"]))
    assert list(res["newact"]) == inst["newact"] and list(res["fv"]) == inst["fv"]


@pytest.mark.skipif(not O.available("ref", "dd"), reason="oracle/_ref only exists where /root/reference does")
@pytest.mark.parametrize("dtype", [O.F64, O.F32])
def test_lbfgsb_restatement_equals_reference_build(dtype):
    ref, port = O.Oracle("ref", "dd"), _port()
    dt = O.NPDT[dtype]
    for n, m, iters, bound in ((400, 4, 25, 1.0), (3001, 7, 18, 0.5)):
        a, b = O.quad_problem(n, 30.0, 5, dtype)
        lb, ub = -bound * np.ones(n, dt), bound * np.ones(n, dt)
        x0 = np.linspace(-2, 2, n).astype(dt)  # starts outside the box: exercises force_bounds
        p = O.lbfgsb_params(m=m, epsilon=0, epsilon_rel=0, max_iterations=iters)  # default past = 1, delta = 1e-10
        t1, t2 = O.TraceBuf(n, cap=600), O.TraceBuf(n, cap=600)
        x1, r1 = ref.lbfgsb(dtype, O.OBJ_QUAD, x0, lb, ub, p, a=a, b=b, trace=t1)
        x2, r2 = port.lbfgsb(dtype, O.OBJ_QUAD, x0, lb, ub, p, a=a, b=b, trace=t2)
        assert (r1.niter, r1.nfev, r1.status, r1.msg) == (r2.niter, r2.nfev, r2.status, r2.msg)
        assert np.array_equal(x1, x2) and np.array_equal(t1.xs[:t1.count], t2.xs[:t2.count])
    # mixed infinite bounds on the Rosenbrock objective
    n = 600
    x0 = O.rosen_x0(n, 3, dtype)
    lb = np.where(np.arange(n) % 3 == 0, -np.inf, -0.5).astype(dt)
    ub = np.where(np.arange(n) % 5 == 0, np.inf, 0.9).astype(dt)
    p = O.lbfgsb_params(m=5, max_iterations=40)
    x1, r1 = ref.lbfgsb(dtype, O.OBJ_ROSEN, x0, lb, ub, p)
    x2, r2 = port.lbfgsb(dtype, O.OBJ_ROSEN, x0, lb, ub, p)
    assert (r1.niter, r1.nfev, r1.status) == (r2.niter, r2.nfev, r2.status) and np.array_equal(x1, x2)


def test_hessian_fixture_is_consistent_and_reproducible():
    """tests/golden/hessian_golden.json: B symmetric, B*H = I; oracle/_ref reproduces it bit for bit when present."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hessian_golden.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 3
    for c in cases:
        n = c["n"]
        B = np.array([float.fromhex(v) for v in c["B"]]).reshape(n, n, order="F")
        H = np.array([float.fromhex(v) for v in c["H"]]).reshape(n, n, order="F")
        assert np.abs(B - B.T).max() <= 1e-12 * np.abs(B).max()
        assert np.abs(B @ H - np.eye(n)).max() < 1e-9
    try:
        ref = O.Oracle("ref", "dd")
    except OSError:
        return
    c = cases[0]
    x, r, B, H = ref.lbfgs_hessians(O.F64, c["ls"], c["obj"], np.zeros(c["n"]),
                                    O.lbfgs_params(m=c["m"], max_iterations=c["max_iterations"], epsilon=1e-6))
    assert r.niter == c["niter"]
    assert [float(v).hex() for v in B.ravel(order="F")] == c["B"]


@pytest.mark.parametrize("ls,obj", [(O.LS_MT, O.OBJ_ROSEN), (O.LS_NW, O.OBJ_QUAD), (O.LS_BT, O.OBJ_ROSEN)])
@pytest.mark.parametrize("R", [2, 64])
def test_replicated_problem_mode_is_the_exact_image_of_the_tiled_problem(ls, obj, R):
    """The knob behind the full-size GPU parity tests (tests/test_full_size_gpu.py): the restatement on a base problem
    of size p with every n-length sum multiplied by R == the R-fold tiled problem, bit for bit -- checked here against the
    tiled problem solved by the UNMODIFIED reference headers (oracle/_ref) when present, else by the restatement."""
    if not O.available("port", "dd"):
        pytest.skip("restatement not built")
    port = O.Oracle("port", "dd")
    big_oracle = O.Oracle("ref", "dd") if O.available("ref", "dd") else port
    p = 250
    par = O.lbfgs_params(m=4, epsilon=0, epsilon_rel=0, max_iterations=14)
    if obj == O.OBJ_ROSEN:
        x0, a, b = O.rosen_x0(p), None, None
    else:
        a, b = O.quad_problem(p)
        x0 = np.zeros(p)
    tile = lambda v: None if v is None else np.tile(v, R)
    trb, trs = O.TraceBuf(p * R, cap=64, with_x=False), O.TraceBuf(p, cap=64, with_x=False)
    xb, rb = big_oracle.lbfgs(O.F64, ls, obj, tile(x0), par, a=tile(a), b=tile(b), trace=trb)
    port.set_replication(R)
    try:
        xs, rs = port.lbfgs(O.F64, ls, obj, x0, par, a=a, b=b, trace=trs)
    finally:
        port.set_replication(1)
    assert (rb.niter, rb.nfev, rb.fx, rb.gnorm) == (rs.niter, rs.nfev, rs.fx, rs.gnorm)
    assert np.array_equal(trb.fx[:trb.count], trs.fx[:trs.count])
    assert np.array_equal(xb.reshape(R, p), np.tile(xs, (R, 1)))
