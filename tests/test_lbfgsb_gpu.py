"""-m gpu: L-BFGS-B device path (Cauchy point, subspace minimisation, driver) against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

import golden_util as G

pytestmark = pytest.mark.gpu
GOLD = G.load("lbfgsb_golden.json")


@pytest.fixture(scope="module")
def boracle(oracle):
    if not oracle.supports_lbfgsb:
        pytest.skip("this oracle build has no L-BFGS-B entry points; golden fixtures cover the path")
    return oracle


@pytest.fixture(scope="module")
def A():
    import lbfgspp_amd as A
    core, _ = A.load()
    assert core.lbfgsx_device_count() >= 1
    return A


def _device_cauchy_subspace(A, dtype, m, S, Y, x0, g, lb, ub, max_submin=10, subspace=True):
    _, sol = A.load()
    dt = O.NPDT[dtype]
    n = x0.size
    S = np.ascontiguousarray(S, dt).reshape(-1, n)
    Y = np.ascontiguousarray(Y, dt).reshape(-1, n)
    npairs = S.shape[0]
    xcp = np.empty(n, dt)
    drt = np.empty(n, dt) if subspace else None
    vecc = np.zeros(2 * m, np.float64)
    state = np.zeros(n, np.uint8)
    counts = (C.c_longlong * 4)()
    err = C.create_string_buffer(256)
    f = sol.lbfgsx_test_cauchy_subspace
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int64, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_void_p,
                                                                               C.c_void_p, C.c_void_p, C.c_void_p,
                                                                               C.c_char_p, C.c_int]
    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
    args = [np.ascontiguousarray(v, dt) for v in (x0, g, lb, ub)]
    rc = f(dtype, n, m, npairs, p(S), p(Y), p(args[0]), p(args[1]), p(args[2]), p(args[3]), max_submin, p(xcp), p(vecc),
           p(state), p(drt), C.cast(counts, C.c_void_p), err, 256)
    assert rc == 0, err.value
    return dict(xcp=xcp, vecc=vecc[:2 * min(npairs, m)], state=state, drt=drt, nact=counts[0], nfree=counts[1],
                crossings=counts[2], sweeps=counts[3])


def _instance(rng, n, npairs, dtype, mode):
    dt = O.NPDT[dtype]
    S = rng.standard_normal((max(npairs, 1), n))[:npairs]
    Y = S * (1.0 + rng.random((npairs, n))) + 0.05 * rng.standard_normal((npairs, n))
    lb = -1.0 - rng.random(n)
    ub = 1.0 + rng.random(n)
    x0 = np.clip(rng.standard_normal(n), lb, ub)
    g = rng.standard_normal(n) * (10.0 if mode != "gentle" else 0.3)
    if mode == "edge":
        fixed = rng.random(n) < 0.05
        ub[fixed] = lb[fixed]
        x0[fixed] = lb[fixed]
        g[rng.random(n) < 0.05] = 0.0                # free forever (brk = inf)
        onb = rng.random(n) < 0.05
        x0[onb] = ub[onb]                            # start on a bound
        tie = rng.random(n) < 0.1                    # a tie group of equal break points
        x0[tie], lb[tie], ub[tie], g[tie] = 0.0, -1.0, 1.0, 4.0
    return (S.astype(dt), Y.astype(dt), x0.astype(dt), g.astype(dt), lb.astype(dt), ub.astype(dt))


@pytest.mark.parametrize("dtype", [O.F64])
@pytest.mark.parametrize("n,m,npairs,mode", [(3000, 6, 0, "hard"), (3000, 6, 4, "hard"), (5000, 6, 9, "edge"),
                                             (4096, 8, 8, "gentle"), (2500, 5, 5, "edge"), (64, 3, 2, "hard")])
def test_cauchy_and_subspace_match_oracle(A, boracle, dtype, n, m, npairs, mode):
    rng = np.random.default_rng(100 + n + npairs)
    S, Y, x0, g, lb, ub = _instance(rng, n, npairs, dtype, mode)
    ref = boracle.cauchy_subspace(dtype, m, S, Y, x0, g, lb, ub, max_submin=10)
    got = _device_cauchy_subspace(A, dtype, m, S, Y, x0, g, lb, ub, max_submin=10)
    # identical index sets (the device keeps them as a state byte)
    newact = np.zeros(n, bool)
    newact[ref["newact"]] = True
    free = np.zeros(n, bool)
    free[ref["fv"]] = True
    assert got["nact"] == newact.sum() and got["nfree"] == free.sum()
    assert np.array_equal((got["state"] & 2) != 0, newact)
    assert np.array_equal((got["state"] & 1) != 0, free)
    scale = max(1.0, np.abs(ref["xcp"]).max())
    assert np.abs(got["xcp"] - ref["xcp"]).max() <= 1e-12 * scale
    if ref["vecc"].size:
        assert np.abs(got["vecc"] - ref["vecc"]).max() <= 1e-11 * max(1.0, np.abs(ref["vecc"]).max())
    dscale = max(1.0, np.abs(ref["drt"]).max())
    assert np.abs(got["drt"] - ref["drt"]).max() <= 1e-9 * dscale


def test_cauchy_all_on_bounds(A):
    """nfree < 1 and nord < 1: xcp = x0, empty sets (reference Cauchy.h:140-145)."""
    n, m = 100, 4
    x0 = np.ones(n)
    lb, ub = np.ones(n), np.ones(n)
    g = np.linspace(-1, 1, n)
    got = _device_cauchy_subspace(A, O.F64, m, np.zeros((0, n)), np.zeros((0, n)), x0, g, lb, ub)
    assert got["nact"] == 0 and got["nfree"] == 0 and np.array_equal(got["xcp"], x0)
    assert np.array_equal(got["drt"], np.zeros(n))


def _traj(A, oracle, n, m, iters, kappa=10.0, dtype=O.F64, bound=1.0):
    a, b = O.quad_problem(n, kappa, 1, dtype)
    dt = O.NPDT[dtype]
    lb, ub = -bound * np.ones(n, dt), bound * np.ones(n, dt)
    p = O.lbfgsb_params(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters)
    tr_ref = O.TraceBuf(n, cap=1024)
    x_ref, r_ref = oracle.lbfgsb(dtype, O.OBJ_QUAD, np.zeros(n, dt), lb, ub, p, a=a, b=b, trace=tr_ref)
    s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters), dtype=dt)
    tr = A.TraceBuffer(n, cap=1024)
    x = np.zeros(n, dt)
    niter, fx = s.minimize(A.DiagQuadratic(a, b), x, lb, ub, trace=tr)
    return dict(x=x, niter=niter, fx=fx, nfev=s.last.nfev, tr=tr, x_ref=x_ref, r_ref=r_ref, tr_ref=tr_ref, stats=s.stats())


@pytest.mark.parametrize("n,m,iters", [(2000, 6, 15), (20000, 10, 25)])
def test_trajectory_box_quadratic_f64(A, boracle, n, m, iters, tol=1e-10):
    r = _traj(A, boracle, n, m, iters)
    assert (r["niter"], r["nfev"]) == (r["r_ref"].niter, r["r_ref"].nfev)
    k = r["tr_ref"].count
    assert r["tr"].count == k
    err = np.abs(r["tr"].xs[:k] - r["tr_ref"].xs[:k]).max()
    assert err <= tol, "iterates deviate by %.3g" % err
    assert np.abs(r["x"] - r["x_ref"]).max() <= tol
    # same active set at the end
    assert np.array_equal(np.abs(r["x"]) == 1.0, np.abs(r["x_ref"]) == 1.0)
    assert abs(r["fx"] - r["r_ref"].fx) <= 1e-12 * abs(r["r_ref"].fx)


def test_box_quadratic_converges_to_projected_solution(A):
    """property check without the oracle: the minimiser of the separable box QP is clip(b/a, lb, ub)."""
    n = 50000
    a, b = O.quad_problem(n, 10.0, 1)
    lb, ub = -np.ones(n), np.ones(n)
    s = A.LBFGSBSolver(A.LBFGSBParam(m=10, epsilon=1e-5, epsilon_rel=0.0, past=0, max_iterations=400))
    x = np.zeros(n)
    niter, fx = s.minimize(A.DiagQuadratic(a, b), x, lb, ub)
    assert niter < 400
    assert np.abs(x - np.clip(b / a, lb, ub)).max() < 1e-4
    assert s.final_grad_norm() <= 1e-5


def test_lbfgsb_argument_errors(A):
    s = A.LBFGSBSolver(A.LBFGSBParam())
    with pytest.raises(ValueError, match="'lb' and 'ub' must have the same size as 'x'"):
        s.minimize(A.DiagQuadratic(np.ones(4), np.ones(4)), np.zeros(4), -np.ones(3), np.ones(4))
    with pytest.raises(ValueError, match="'max_submin' must be non-negative"):
        A.LBFGSBSolver(A.LBFGSBParam(max_submin=-1))
    # the reference has no limit on m; this implementation's masked operators stop at 2m = 80 and say so
    s = A.LBFGSBSolver(A.LBFGSBParam(m=41))
    with pytest.raises(ValueError, match="m <= 40"):
        s.minimize(A.DiagQuadratic(np.ones(8), np.ones(8)), np.zeros(8), -np.ones(8), np.ones(8))
    s = A.LBFGSBSolver(A.LBFGSBParam(m=40, max_iterations=3))
    s.minimize(A.DiagQuadratic(np.ones(8), np.ones(8)), np.zeros(8), -np.ones(8), np.ones(8))
    # ... and the break-point list carries 32-bit indices, as the reference's index sets do
    s = A.LBFGSBSolver(A.LBFGSBParam(m=2), dtype=np.float32)
    with pytest.raises(ValueError, match="n < 2\\^31"):
        s.prepare(2 ** 31)


_INF = float("inf")


@pytest.mark.parametrize("name,lb,ub,x0", [
    ("n1", [-0.25], [0.5], [0.4]),
    ("n1_fixed", [0.3], [0.3], [0.3]),                                   # lb == ub: nothing is free, the search ends at once
    ("n2_unbounded", [-_INF, -_INF], [_INF, _INF], [0.0, 0.0]),          # no finite bound: the L-BFGS-B machinery on a free problem
    ("n3_one_fixed", [-1.0, 0.2, -1.0], [1.0, 0.2, 1.0], [0.0, 0.2, 0.5]),
    ("n7_half_lines", [-_INF, -0.1, -_INF, 0.0, -2.0, -_INF, 0.05], [0.1, _INF, _INF, 0.0, _INF, -0.3, 0.06],
     [0.0, 0.0, 3.0, 0.0, -1.0, -0.5, 0.055]),
    ("n5_start_outside", [-0.5] * 5, [0.5] * 5, [-3.0, 3.0, 0.0, 0.7, -0.7]),
])
@pytest.mark.parametrize("dtype", [O.F64, O.F32])
def test_smallest_problems_and_infinite_bounds_match_the_reference(A, boracle, name, lb, ub, x0, dtype):
    """n = 1 ... 7 with bounds at +-inf, lb == ub and a start outside the box (LBFGSB.h:55-66 force_bounds, Cauchy.h:93-133 break
    points of half-lines): same outcome as the reference's solver -- counts, minimiser, or the same failure"""
    dt = O.NPDT[dtype]
    n = len(lb)
    rng = np.random.default_rng(n)
    a = (1.0 + 9.0 * rng.random(n)).astype(dt)
    b = (4.0 * rng.random(n) - 2.0).astype(dt)
    lb, ub, x0 = np.array(lb, dt), np.array(ub, dt), np.array(x0, dt)
    kw = dict(m=4, epsilon=1e-6 if dtype == O.F64 else 1e-4, epsilon_rel=0, past=0, max_iterations=40)
    x_ref, r_ref = boracle.lbfgsb(dtype, O.OBJ_QUAD, x0, lb, ub, O.lbfgsb_params(**kw), a=a, b=b)
    s = A.LBFGSBSolver(A.LBFGSBParam(**kw), dtype=dt)
    x = x0.copy()
    try:
        niter, fx = s.minimize(A.DiagQuadratic(a, b), x, lb, ub)
        status = 0
    except (RuntimeError, ArithmeticError, ValueError):
        status, niter, fx = s.last.status, s.last.niter, s.last.fx
    assert (status == 0) == (r_ref.status == 0), (name, status, r_ref.status, r_ref.msg)
    assert (niter, s.last.nfev) == (r_ref.niter, r_ref.nfev), name
    tol = 1e-10 if dtype == O.F64 else 1e-5
    assert np.abs(x - x_ref).max() <= tol, name
    if status == 0:
        assert np.all(x >= lb) and np.all(x <= ub)
        assert np.abs(x - np.clip(b / a, lb, ub)).max() <= (1e-5 if dtype == O.F64 else 1e-3)


@pytest.mark.parametrize("m,iters", [(20, 30), (22, 34), (24, 36), (27, 40), (30, 45), (36, 45), (40, 50)])
def test_lbfgsb_long_histories_match_oracle(A, boracle, m, iters):
    """2c beyond the single-launch widths (multi-dot chunks of 8 columns, blocked Gram, host Cauchy search) up to the
    limit 2m = 80 of the masked operators: same iteration / evaluation counts and iterates as the reference."""
    n = 3000
    a, b = O.quad_problem(n, 200.0, 3, O.F64)
    lb, ub = -0.4 * np.ones(n), 0.6 * np.ones(n)
    p = O.lbfgsb_params(m=m, max_iterations=iters, epsilon=0, epsilon_rel=0, past=0)
    x_ref, r_ref = boracle.lbfgsb(O.F64, O.OBJ_QUAD, np.zeros(n), lb, ub, p, a=a, b=b)
    s = A.LBFGSBSolver(A.LBFGSBParam(m=m, max_iterations=iters, epsilon=0, epsilon_rel=0, past=0))
    x = np.zeros(n)
    niter, fx = s.minimize(A.DiagQuadratic(a, b), x, lb, ub)
    assert niter == r_ref.niter == iters and s.last.nfev == r_ref.nfev
    assert np.abs(x - x_ref).max() <= 1e-10


def test_lbfgsb_start_outside_bounds_is_projected(A, boracle):
    oracle = boracle
    n = 3000
    a, b = O.quad_problem(n)
    lb, ub = -0.5 * np.ones(n), 0.5 * np.ones(n)
    x0 = np.linspace(-3, 3, n)
    p = O.lbfgsb_params(m=5, max_iterations=12, epsilon=0, epsilon_rel=0, past=0)
    x_ref, r_ref = oracle.lbfgsb(O.F64, O.OBJ_QUAD, x0, lb, ub, p, a=a, b=b)
    s = A.LBFGSBSolver(A.LBFGSBParam(m=5, max_iterations=12, epsilon=0, epsilon_rel=0, past=0))
    x = x0.copy()
    niter, fx = s.minimize(A.DiagQuadratic(a, b), x, lb, ub)
    assert niter == r_ref.niter and s.last.nfev == r_ref.nfev
    assert np.abs(x - x_ref).max() <= 1e-10


# ---------------------------------------------------------------- golden fixtures (need no oracle library)
def _golden_instance(seed, n, npairs, mode):
    rng = np.random.default_rng(seed)
    S = rng.standard_normal((max(npairs, 1), n))[:npairs]
    Y = S * (1.0 + rng.random((npairs, n))) + 0.05 * rng.standard_normal((npairs, n))
    lb = -1.0 - rng.random(n)
    ub = 1.0 + rng.random(n)
    x0 = np.clip(rng.standard_normal(n), lb, ub)
    g = rng.standard_normal(n) * (10.0 if mode != "gentle" else 0.3)
    if mode == "edge":
        fixed = rng.random(n) < 0.05
        ub[fixed] = lb[fixed]
        x0[fixed] = lb[fixed]
        g[rng.random(n) < 0.05] = 0.0
        onb = rng.random(n) < 0.05
        x0[onb] = ub[onb]
        tie = rng.random(n) < 0.1
        x0[tie], lb[tie], ub[tie], g[tie] = 0.0, -1.0, 1.0, 4.0
    return S, Y, x0, g, lb, ub


@pytest.mark.parametrize("inst", GOLD["instances"], ids=["seed%d" % i["seed"] for i in GOLD["instances"]])
def test_cauchy_subspace_golden(A, inst):
    n, m = inst["n"], inst["m"]
    S, Y, x0, g, lb, ub = _golden_instance(inst["seed"], n, inst["npairs"], inst["mode"])
    got = _device_cauchy_subspace(A, O.F64, m, S, Y, x0, g, lb, ub, max_submin=10)
    newact = np.zeros(n, bool)
    newact[inst["newact"]] = True
    free = np.zeros(n, bool)
    free[inst["fv"]] = True
    assert np.array_equal((got["state"] & 2) != 0, newact) and np.array_equal((got["state"] & 1) != 0, free)
    xcp, drt, vecc = G.unhex(inst["xcp"]), G.unhex(inst["drt"]), G.unhex(inst["vecc"])
    assert np.abs(got["xcp"] - xcp).max() <= 1e-12 * max(1.0, np.abs(xcp).max())
    if vecc.size:
        assert np.abs(got["vecc"] - vecc).max() <= 1e-11 * max(1.0, np.abs(vecc).max())
    assert np.abs(got["drt"] - drt).max() <= 1e-9 * max(1.0, np.abs(drt).max())


@pytest.mark.parametrize("case", GOLD["trajectories"], ids=[c["name"] for c in GOLD["trajectories"]])
def test_lbfgsb_trajectory_golden(A, case, tol=1e-10):
    n, m = case["n"], case["m"]
    a, b = O.quad_problem(n)
    s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=case["max_iterations"]))
    tr = A.TraceBuffer(n, cap=1024, stride=case["stride"])
    x = np.zeros(n)
    niter, fx = s.minimize(A.DiagQuadratic(a, b), x, -np.ones(n), np.ones(n), trace=tr)
    assert (niter, s.last.nfev) == (case["niter"], case["nfev"])
    k = tr.count
    xs = G.unhex(case["trace_xs"]).reshape(k, -1)
    assert np.abs(tr.xs[:k] - xs).max() <= tol
    assert np.abs(x[::case["stride"]] - G.unhex(case["x_sample"])).max() <= tol
    assert abs(fx - float.fromhex(case["fx"])) <= 1e-12 * abs(fx)


@pytest.mark.rhs_pass
@pytest.mark.parametrize("max_submin", [10, 2, 1])
def test_fused_subspace_passes_are_bit_identical_to_the_unfused_sequence(A, monkeypatch, max_submin):
    """Default path (one-pass Gram with the rhs / linear-term prologue, solve fused with W_F'y, one-launch
    multi-dot, one element-wise pass between two BOXCQP solves, add_correction's dots taken by the W'd pass) against the statement-by-statement sequence
    (LBFGSX_GRAM=blocked, LBFGSX_MULTIDOT=chunked, LBFGSX_SUB_FUSE=0, LBFGSX_CORR_DEFER=0): the fusions only remove passes, so every iterate
    must agree to the last bit -- also when the sweeps run out (max_submin = 2, 1: the fallback ladder of
    SubspaceMin.h:276-296)."""
    n, m, iters = 30000, 8, 18
    a, b = O.quad_problem(n, 30.0, 3, O.F64)
    res = {}
    for label, env in (("fused", {}), ("unfused", {"LBFGSX_GRAM": "blocked", "LBFGSX_MULTIDOT": "chunked",
                                                     "LBFGSX_SUB_FUSE": "0", "LBFGSX_CORR_DEFER": "0"})):
        for k in ("LBFGSX_GRAM", "LBFGSX_MULTIDOT", "LBFGSX_SUB_FUSE", "LBFGSX_CORR_DEFER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters,
                                          max_submin=max_submin))
        tr = A.TraceBuffer(n, cap=256, stride=7)
        x = np.zeros(n)
        niter, fx = s.minimize(A.DiagQuadratic(a, b), x, -0.7 * np.ones(n), 0.9 * np.ones(n), trace=tr)
        res[label] = (niter, s.last.nfev, fx, x.copy(), tr.xs[:tr.count].copy(), s.stats()["submin_sweeps"],
                      s.stats()["submin_unconverged"])
    f, u = res["fused"], res["unfused"]
    assert f[:3] == u[:3] and f[5] == u[5] and f[5] > 0
    assert np.array_equal(f[3], u[3]) and np.array_equal(f[4], u[4])
    assert f[6] == u[6] and (max_submin >= 10 or f[6] > 0)


@pytest.mark.rhs_pass
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("m,max_submin", [(8, 10), (10, 3), (3, 10), (14, 10), (12, 10), (20, 10), (40, 4)])
def test_sweep_statements_riding_on_the_solve_change_no_bit(A, monkeypatch, m, max_submin, dtype):
    """lbfgsx_b_solve_sweep / lbfgsx_b_lu_sweep (the sweep's element-wise statements inside the solve's pass, the rows of
    L and U through the index list) against the separate passes (LBFGSX_SWEEP_SOLVE_FUSE=0): same statements on the same
    values, so the same trajectory bit for bit and the same sweep counts; the fused form must actually have run."""
    n, iters = 40003, max(20, m + 8)
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    a, b = O.quad_problem(n, 30.0, 5, dt)
    res = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("LBFGSX_SWEEP_SOLVE_FUSE", fuse)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters, max_submin=max_submin),
                           dtype=npdt)
        tr = A.TraceBuffer(n, cap=256, stride=13)
        x = np.zeros(n, dtype=npdt)
        try:
            niter, fx = s.minimize(A.DiagQuadratic(a, b), x, (-0.7 * np.ones(n)).astype(npdt), (0.9 * np.ones(n)).astype(npdt),
                                   trace=tr)
        except RuntimeError:  # f32 may stop on a line-search failure: both forms must then stop at the same place
            niter, fx = -1, float("nan")
        st = s.stats()
        res[fuse] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), st["submin_sweeps"], st["submin_unconverged"],
                     st["submin_fused_sweeps"])
    f, u = res["1"], res["0"]
    assert f[:2] == u[:2] and f[4:6] == u[4:6] and f[4] > 0
    assert np.array_equal(f[2], u[2]) and np.array_equal(f[3], u[3])
    assert u[6] == 0 and f[6] > 0


@pytest.mark.rhs_pass
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n,m,max_submin", [(70001, 8, 10), (70001, 10, 2), (65536, 3, 10), (90000, 14, 10), (70001, 12, 10),
                                            (90000, 20, 10), (65536, 40, 10)])
def test_compact_copy_of_the_free_rows_changes_no_bit(A, monkeypatch, n, m, max_submin, dtype):
    """The passes of the BOXCQP sweeps reading the compact copy of the free rows that the first solve's Gram pass leaves
    (lbfgsx_b_set_compaction) against the same passes reading all n rows through the state-byte mask
    (LBFGSX_COMPACT_FREE=0): the same rows in the same order, so the same trajectory bit for bit."""
    iters = max(20, m + 8)
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    a, b = O.quad_problem(n, 30.0, 9, dt)
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("LBFGSX_COMPACT_FREE", on)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters, max_submin=max_submin),
                           dtype=npdt)
        tr = A.TraceBuffer(n, cap=256, stride=17)
        x = np.zeros(n, dtype=npdt)
        try:
            niter, fx = s.minimize(A.DiagQuadratic(a, b), x, (-0.7 * np.ones(n)).astype(npdt), (0.9 * np.ones(n)).astype(npdt),
                                   trace=tr)
        except RuntimeError:
            niter, fx = -1, float("nan")
        st = s.stats()
        res[on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), st["submin_sweeps"], st["submin_unconverged"])
    f, u = res["1"], res["0"]
    assert f[:2] == u[:2] and f[4:] == u[4:] and f[4] > 0
    assert np.array_equal(f[2], u[2]) and np.array_equal(f[3], u[3])


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n,m,max_submin,leave", [(70001, 8, 10, False), (70001, 10, 2, False), (65536, 3, 10, False),
                                                 (90000, 14, 10, False), (120000, 10, 10, True), (70001, 12, 10, False),
                                                 (90000, 20, 10, False), (65536, 40, 10, False), (90000, 20, 10, True)])
def test_compact_vectors_of_the_free_rows_change_no_bit(A, monkeypatch, n, m, max_submin, leave, dtype):
    """While the BOXCQP sweeps walk the compact copy, vecy / yfallback / lambda / mu / rhs / c_F / l - x0 / u - x0 and the
    partition bits of the free rows sit at the rows' positions (lbfgsx_b_compact_vec_counts) instead of at the rows
    (LBFGSX_COMPACT_VEC=0): the same statements on the same values, so the same trajectory bit for bit and the same sweep
    counts; the compact form must have run.  Without the complement identity (LBFGSX_GRAM_COMPLEMENT=0) the second solve
    of a minimisation is a masked Gram pass over the rows: the vectors go back to their rows in the middle of the
    minimisation -- also without a changed bit."""
    import ctypes as C
    from lbfgspp_amd import _lib
    core, _ = _lib.load()
    iters = max(25, m + 10)
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    a, b = O.quad_problem(n, 30.0, 9, dt)
    if leave:
        monkeypatch.setenv("LBFGSX_GRAM_COMPLEMENT", "0")
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("LBFGSX_COMPACT_VEC", on)
        core.lbfgsx_b_compact_vec_counts(None, 1)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters, max_submin=max_submin),
                           dtype=npdt)
        tr = A.TraceBuffer(n, cap=256, stride=17)
        x = np.zeros(n, dtype=npdt)
        try:
            niter, fx = s.minimize(A.DiagQuadratic(a, b), x, (-0.7 * np.ones(n)).astype(npdt), (0.9 * np.ones(n)).astype(npdt),
                                   trace=tr)
        except RuntimeError:
            niter, fx = -1, float("nan")
        st = s.stats()
        cnt = (C.c_int64 * 4)()
        core.lbfgsx_b_compact_vec_counts(C.byref(cnt), 0)
        res[on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), st["submin_sweeps"], st["submin_unconverged"],
                   st["submin_fused_sweeps"], (cnt[0], cnt[1]))
    f, u = res["1"], res["0"]
    assert f[:2] == u[:2] and f[4:7] == u[4:7] and f[4] > 0
    assert np.array_equal(f[2], u[2]) and np.array_equal(f[3], u[3])
    assert u[7] == (0, 0) and f[7][0] > 0
    if leave:
        assert f[7][1] > 0
    else:
        assert f[7][1] <= f[7][0]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n,m,iters,cap", [(70001, 8, 40, None), (200000, 10, 50, None), (65536, 5, 30, None), (120000, 10, 40, "8")])
def test_partial_sort_candidates_listed_by_the_cauchy_build_change_no_bit(A, monkeypatch, n, m, iters, cap, dtype):
    """The break points <= tau that the partial sort orders: listed by the Cauchy build as it meets them, the list put in
    row order, then the stable sort by break point (lbfgsx_b_cauchy_build_partial) -- against rocprim::select's ordered
    compaction in a pass of its own (LBFGSX_SELECT_INLINE=0).  The same list in the same order: the same bits.  With room
    for 8 candidates (LBFGSX_SELECT_CAP) the list overflows and those searches take the separate pass."""
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    a, b = O.quad_problem(n, 30.0, 9, dt)
    if cap:
        monkeypatch.setenv("LBFGSX_SELECT_CAP", cap)
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("LBFGSX_SELECT_INLINE", on)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters, max_submin=10), dtype=npdt)
        tr = A.TraceBuffer(n, cap=256, stride=17)
        x = np.zeros(n, dtype=npdt)
        try:
            niter, fx = s.minimize(A.DiagQuadratic(a, b), x, (-0.7 * np.ones(n)).astype(npdt), (0.9 * np.ones(n)).astype(npdt),
                                   trace=tr)
        except RuntimeError:
            niter, fx = -1, float("nan")
        st = s.stats()
        res[on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), st["gcp_partial_sorts"], st["gcp_sorted"])
    f, u = res["1"], res["0"]
    assert f[:2] == u[:2] and f[4:] == u[4:] and f[4] > 0
    assert np.array_equal(f[2], u[2]) and np.array_equal(f[3], u[3])


def test_polled_completion_serves_the_waits_and_changes_no_bit(A, monkeypatch):
    """The kernels whose results the host reads next end with a sequence number stored in host-mapped memory after the
    results; the host polls that word instead of waiting for the stream (ctx.hpp: poll_arm / poll_wait).  Against
    LBFGSX_POLL=0 (every wait is a stream wait): the same trajectory bit for bit; most waits are served by polling and
    none of them runs into the time-out (which would mean a kernel that was armed and never signalled)."""
    import ctypes as C
    from lbfgspp_amd import _lib as L
    core, _ = L.load()
    n, iters = 300000, 40
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("LBFGSX_POLL", on)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=10, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters))
        ctx = s.prepare(n)
        L.check(core.lbfgsx_gen_diag_quad(ctx, 10.0, 1))
        L.check(core.lbfgsx_fill(ctx, L.VEC_X, 0.0))
        L.check(core.lbfgsx_fill(ctx, L.VEC_LB, -1.0))
        L.check(core.lbfgsx_fill(ctx, L.VEC_UB, 1.0))
        niter, fx = s.minimize_resident(A.DiagQuadratic(), n)
        x = np.empty(n)
        L.check(core.lbfgsx_download(ctx, L.VEC_X, x.ctypes.data_as(C.c_void_p)))
        pc = (C.c_int64 * 2)()
        L.check(core.lbfgsx_poll_counts(ctx, C.byref(pc)))
        res[on] = (niter, s.last.nfev, fx, x, int(pc[0]), int(pc[1]), s.stats()["submin_sweeps"])
        s.close()
    f, u = res["1"], res["0"]
    assert f[:3] == u[:3] and f[6] == u[6] and np.array_equal(f[3], u[3])
    assert u[4] == 0 and u[5] == 0
    assert f[4] > 10 * iters and f[5] == 0


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n,m,iters", [(70001, 8, 40), (70001, 10, 45), (65536, 3, 30), (200000, 10, 60), (70001, 12, 40),
                                       (90000, 20, 60), (65536, 40, 100)])
def test_grams_launched_ahead_of_their_request_change_no_bit(A, monkeypatch, n, m, iters, dtype):
    """The Gram over the rows of L u U rides behind the W_L'l / W_U'u pass, the Grams over the rows that entered / left the
    free set behind the selected-entries pass of the carried first solve: the same kernels on the same data, launched
    earlier and fetched with the wait of the pass before them (one host round trip less each).  Against launching them when
    they are asked for (LBFGSX_SYNC_MERGE=0): the same bits; and the early launches must have been used."""
    import ctypes as C
    from lbfgspp_amd import _lib
    core, _ = _lib.load()
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    a, b = O.quad_problem(n, 30.0, 9, dt)
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("LBFGSX_SYNC_MERGE", on)
        core.lbfgsx_b_compact_vec_counts(None, 1)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters, max_submin=10), dtype=npdt)
        tr = A.TraceBuffer(n, cap=256, stride=17)
        x = np.zeros(n, dtype=npdt)
        try:
            niter, fx = s.minimize(A.DiagQuadratic(a, b), x, (-0.7 * np.ones(n)).astype(npdt), (0.9 * np.ones(n)).astype(npdt),
                                   trace=tr)
        except RuntimeError:
            niter, fx = -1, float("nan")
        st = s.stats()
        cnt = (C.c_int64 * 4)()
        core.lbfgsx_b_compact_vec_counts(C.byref(cnt), 0)
        res[on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), st["submin_sweeps"], cnt[3])
    f, u = res["1"], res["0"]
    assert f[:2] == u[:2] and f[4] == u[4]
    assert np.array_equal(f[2], u[2]) and np.array_equal(f[3], u[3])
    assert u[5] == 0 and f[5] > 0


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n,m,iters,cap", [(70001, 12, 40, None), (90000, 20, 60, None), (65536, 40, 100, None),
                                           (70001, 8, 40, None), (70001, 10, 45, None), (65536, 5, 30, None), (200000, 10, 60, None),
                                           (120000, 10, 40, "8")])
def test_cauchy_dots_from_the_kept_compact_copy_change_no_bit(A, monkeypatch, n, m, iters, cap, dtype):
    """p = W'd of the Cauchy search and the deferred dots of add_correction as sums over the positions of the compact copy
    kept from the previous iteration plus the listed rows outside it (k_multidot2_wf) against the pass over all n rows
    (LBFGSX_WTD_COMPACT=0): the terms left out are exact zeros and every sum is correctly rounded, so the trajectory is the
    same bit for bit; the compact form must have run.  With a list of 8 rows (LBFGSX_WTD_LIST_CAP) it overflows now and
    then and those iterations take the full pass."""
    import ctypes as C
    from lbfgspp_amd import _lib
    core, _ = _lib.load()
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    a, b = O.quad_problem(n, 30.0, 9, dt)
    if cap:
        monkeypatch.setenv("LBFGSX_WTD_LIST_CAP", cap)
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("LBFGSX_WTD_COMPACT", on)
        core.lbfgsx_b_compact_vec_counts(None, 1)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters, max_submin=10), dtype=npdt)
        tr = A.TraceBuffer(n, cap=256, stride=17)
        x = np.zeros(n, dtype=npdt)
        try:
            niter, fx = s.minimize(A.DiagQuadratic(a, b), x, (-0.7 * np.ones(n)).astype(npdt), (0.9 * np.ones(n)).astype(npdt),
                                   trace=tr)
        except RuntimeError:
            niter, fx = -1, float("nan")
        cnt = (C.c_int64 * 4)()
        core.lbfgsx_b_compact_vec_counts(C.byref(cnt), 0)
        res[on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), cnt[2])
    f, u = res["1"], res["0"]
    assert f[:2] == u[:2]
    assert np.array_equal(f[2], u[2]) and np.array_equal(f[3], u[3])
    assert u[4] == 0 and f[4] > 0


@pytest.mark.rhs_pass
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n,m,iters,age", [(70001, 8, 60, 32), (70001, 10, 45, 32), (65536, 3, 40, 32), (90000, 12, 40, 32),
                                           (200000, 10, 80, 32), (70001, 8, 40, 2), (120000, 10, 40, 5), (65536, 5, 40, 3),
                                           (90000, 20, 60, 32), (65536, 40, 100, 32), (120000, 20, 50, 3), (70001, 10, 330, 256)])
def test_carried_gram_of_the_free_set_changes_no_bit(A, monkeypatch, n, m, iters, age, dtype):
    """W_F'W_F of the first BOXCQP solve from the sums of the previous iteration -- the rows of the two replaced columns
    computed afresh, the other entries corrected by the outer products of the rows that entered or left the free set, all
    in double-double (BFGSMatB::carried_gram) -- against the full Gram pass every iteration (LBFGSX_GRAM_CARRY=0): the
    rounded entries are the same, so the trajectory is bit for bit the same; the carried form must have run -- at every m
    (round 3: m <= 10 only, one lane per entry; the split-row kernel serves 3 (2c + 1) entries for any 2c <= 80) -- across
    its periodic refresh (every 256 iterations by default since round 4 -- 2^-104 per update leaves 2^-96 after 256 -- every
    32 before; both periods and much shorter ones here) and the growth of the history."""
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    a, b = O.quad_problem(n, 30.0, 11, dt)
    monkeypatch.setenv("LBFGSX_GRAM_CARRY_AGE", str(age))  # a short period interleaves full and carried passes differently
    res = {}
    for on in ("1", "0", "nokeep"):
        # "nokeep": carried sums, but the compact copy of the free rows is written afresh every iteration instead of being
        # kept and patched (the rows that entered F appended, the replaced slot's two columns rewritten)
        monkeypatch.setenv("LBFGSX_GRAM_CARRY", "0" if on == "0" else "1")
        monkeypatch.setenv("LBFGSX_COMPACT_KEEP", "0" if on == "nokeep" else "1")
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters), dtype=npdt)
        tr = A.TraceBuffer(n, cap=512, stride=29)
        x = np.zeros(n, dtype=npdt)
        try:
            niter, fx = s.minimize(A.DiagQuadratic(a, b), x, (-0.7 * np.ones(n)).astype(npdt), (0.9 * np.ones(n)).astype(npdt),
                                   trace=tr)
        except RuntimeError:
            niter, fx = -1, float("nan")
        st = s.stats()
        res[on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), st["submin_sweeps"], st["gram_carried"], st["submin_calls"])
    f, u, k = res["1"], res["0"], res["nokeep"]
    assert f[:2] == u[:2] and f[4] == u[4]
    assert np.array_equal(f[2], u[2]) and np.array_equal(f[3], u[3])
    assert k[:2] == u[:2] and k[4] == u[4] and np.array_equal(k[2], u[2]) and np.array_equal(k[3], u[3])
    # without the kept copy a carried pass would have to write a new one, which the selected-entries kernel only does through
    # the round-3 tile form (64 entries: m <= 10); beyond that "nokeep" takes the full pass (same bits either way)
    assert u[5] == 0 and (k[5] == f[5] if m <= 10 else k[5] <= f[5])
    if dtype == "f64":
        assert f[5] >= f[6] // 4, "the carried form ran in %d of %d subspace minimisations" % (f[5], f[6])


@pytest.mark.rhs_pass
@pytest.mark.parametrize("n,m,iters", [(70001, 10, 14), (65536, 20, 24), (65536, 40, 44)])
def test_kept_copy_and_carried_gram_serve_the_iterations_in_which_the_history_fills(A, monkeypatch, n, m, iters):
    """Round 5: the columns of the compact copy are slot-stable (Y slot j in column j, S slot j in column m + j whatever the
    history length), so a copy written at c pairs is the kept copy at c + 1 and the carried Gram -- with the new slot as its one
    fresh pair -- serves the first m iterations too, where until round 4 every new pair moved the S columns and forced a full
    Gram pass and a new copy (at m = 40: 40 of the first 60 iterations).  Same bits as the full pass every iteration, and the
    carried form runs in all but the first few minimisations (no history yet; the first full pass)."""
    a, b = O.quad_problem(n, 30.0, 11, O.F64)
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("LBFGSX_GRAM_CARRY", on)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters))
        tr = A.TraceBuffer(n, cap=512, stride=29)
        x = np.zeros(n)
        niter, fx = s.minimize(A.DiagQuadratic(a, b), x, -0.7 * np.ones(n), 0.9 * np.ones(n), trace=tr)
        st = s.stats()
        res[on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), st["gram_carried"], st["submin_calls"])
    f, u = res["1"], res["0"]
    assert f[:2] == u[:2] and np.array_equal(f[2], u[2]) and np.array_equal(f[3], u[3])
    assert u[4] == 0 and f[5] >= iters - 1
    assert f[4] >= f[5] - 4, "the carried form ran in %d of %d subspace minimisations while the history filled" % (f[4], f[5])


@pytest.mark.parametrize("m", [3, 5, 8, 10, 12, 20, 40])
def test_deferred_correction_dots_change_no_bit(A, monkeypatch, m):
    """add_correction's S's_new / s_new.y_j dots taken by the W'd pass of the following Cauchy search
    (lbfgsx_b_correction_dots_defer, k_multidot2_all for 8 < 2c <= 20) against the pass of their own
    (LBFGSX_CORR_DEFER=0): the same correctly rounded sums, so the same trajectory bit for bit; 2c <= 8 and
    2c > 20 take the separate pass either way."""
    n, iters = 50001, 2 * m + 6  # odd n: the scalar tail of the vectorised pass
    a, b = O.quad_problem(n, 25.0, 7, O.F64)
    res = {}
    for defer in ("1", "0"):
        monkeypatch.setenv("LBFGSX_CORR_DEFER", defer)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters))
        tr = A.TraceBuffer(n, cap=256, stride=11)
        x = np.zeros(n)
        niter, fx = s.minimize(A.DiagQuadratic(a, b), x, -0.6 * np.ones(n), 0.8 * np.ones(n), trace=tr)
        res[defer] = (niter, s.last.nfev, fx, x.copy(), tr.xs[:tr.count].copy())
    f, u = res["1"], res["0"]
    assert f[:3] == u[:3] and f[0] == iters
    assert np.array_equal(f[3], u[3]) and np.array_equal(f[4], u[4])


def test_partial_break_point_sort_is_exact_and_falls_back(A, monkeypatch):
    """lbfgsx_b_cauchy_build_partial: sorting only the break points below tau = factor * (previous Cauchy time) must
    not change a single bit of the trajectory; with a factor < 1 the prefix is regularly too short, which exercises
    the sentinel test and the full re-sort."""
    n, m, iters = 40000, 6, 25
    a, b = O.quad_problem(n, 20.0, 5, O.F64)
    res = {}
    for factor in ("0", "8", "0.4"):
        monkeypatch.setenv("LBFGSX_GCP_TAU_FACTOR", factor)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters))
        x = np.zeros(n)
        niter, fx = s.minimize(A.DiagQuadratic(a, b), x, -0.5 * np.ones(n), 0.8 * np.ones(n))
        st = s.stats()
        res[factor] = (niter, s.last.nfev, fx, x.copy(), st["gcp_crossings"], st["gcp_partial_sorts"], st["gcp_sort_fallbacks"])
    full, part, short = res["0"], res["8"], res["0.4"]
    assert full[5] == 0 and full[6] == 0
    assert part[5] > 0                      # partial sorts were used ...
    assert short[6] > 0                     # ... and redone in full when tau was too small
    for r in (part, short):
        assert r[:3] == full[:3] and r[4] == full[4] and np.array_equal(r[3], full[3])


@pytest.mark.parametrize("n,m,npairs,spread", [(5000, 10, 10, 0), (300001, 6, 4, 6), (1 << 20, 15, 15, 3), (70, 3, 2, 0),
                                               (200003, 10, 7, 12)])
def test_integer_mfma_gram_is_bit_identical_to_the_double_double_gram(A, monkeypatch, n, m, npairs, spread):
    """LBFGSX_GRAM=i8 (csrc/gram_i8.cuh): W_P'W_P on v_mfma_i32_32x32x32_i8 from radix-256 digits of a per-column
    fixed-point grid -- integer sums, one rounding at the end.  Every rounded entry must equal the double-double kernel's
    (both are the correctly rounded exact sum), the un-rounded (hi, lo) pairs must agree to 2^-78 of the column scales, and the v row (kept
    in double-double) is identical.  `spread`: rows scaled by 10^U(-spread, spread) so that most elements sit far below
    their column's maximum (the digits of small elements start many bytes down)."""
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    vp = C.c_void_p
    f = core.lbfgsx_b_gram_fused_dd
    f.restype, f.argtypes = C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    out = {}
    for mode in ("i8", "dd"):
        monkeypatch.setenv("LBFGSX_GRAM", mode)
        h = C.c_void_p()
        L.check(core.lbfgsx_create(C.byref(h), 0, n, m, 0, 1))
        rng = np.random.default_rng(n + 17)
        scale = 10.0 ** rng.uniform(-spread, spread, n) if spread else np.ones(n)
        for k in range(npairs):
            s_ = rng.standard_normal(n) * scale
            y_ = s_ * (1 + rng.random(n))
            L.check(core.lbfgsx_bfgs_add_correction_host(h, s_.ctypes.data_as(vp), y_.ctypes.data_as(vp)))
        d = rng.standard_normal(n) * scale
        L.check(core.lbfgsx_upload(h, L.VEC_D, d.ctypes.data_as(vp)))
        t = 2 * min(npairs, m)
        g, w, gd = np.zeros((t, t)), np.zeros(t), np.zeros(t * (t + 1))
        L.check(f(h, 0, 0, 0, None, None, g.ctypes.data_as(vp), w.ctypes.data_as(vp), gd.ctypes.data_as(vp)))
        core.lbfgsx_destroy(h)
        out[mode] = (g, w, gd.reshape(-1, 2))
    assert np.array_equal(out["i8"][0], out["dd"][0]), "rounded Gram entries differ: max rel %.3g" % (
        np.abs(out["i8"][0] - out["dd"][0]) / np.abs(out["dd"][0])).max()
    assert np.array_equal(out["i8"][1], out["dd"][1])
    hi8, hdd = out["i8"][2], out["dd"][2]
    tot8, totd = hi8[:, 0] + hi8[:, 1], hdd[:, 0] + hdd[:, 1]
    assert np.array_equal(tot8, totd)
    resid = np.abs((hi8[:, 0] - hdd[:, 0]) + (hi8[:, 1] - hdd[:, 1]))
    diag = np.sqrt(np.abs(np.outer(np.diag(out["dd"][0]), np.diag(out["dd"][0]))))
    idx = [(i, j) for i in range(diag.shape[0]) for j in range(i + 1)]
    bound = np.array([diag[i, j] for i, j in idx]) * 2.0 ** -78   # dropped digit pairs: < 2^-82 of the two column maxima per row
    assert (resid <= bound).all(), "un-rounded sums differ by up to 2^%.1f of the column scales" % np.log2((resid / bound).max() * 2.0 ** -78)


@pytest.mark.parametrize("n,m,iters,kappa", [(30000, 8, 18, 30.0), (400000, 10, 12, 10.0)])
def test_integer_mfma_gram_trajectories_change_no_bit(A, monkeypatch, n, m, iters, kappa):
    """whole L-BFGS-B runs (masks, prologues, the complement identity of the BOXCQP sweeps on the un-rounded sums) with
    the integer Gram against the double-double Gram: identical iterates"""
    a, b = O.quad_problem(n, kappa, 3, O.F64)
    lb, ub = -np.ones(n), np.ones(n)
    res = {}
    for mode in ("i8", "dd"):
        monkeypatch.setenv("LBFGSX_GRAM", mode)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters))
        x = np.zeros(n)
        tr = A.TraceBuffer(n, cap=512, stride=max(1, n // 5000))
        niter, fx = s.minimize(A.DiagQuadratic(a, b), x, lb, ub, trace=tr)
        res[mode] = (niter, s.last.nfev, fx, x, tr.xs[:tr.count].copy())
        s.close()
    assert res["i8"][:3] == res["dd"][:3]
    assert np.array_equal(res["i8"][3], res["dd"][3]) and np.array_equal(res["i8"][4], res["dd"][4])


def test_integer_mfma_gram_reads_and_writes_the_compact_copy(A, oracle, monkeypatch):
    """Round 3: the matrix-core Gram (LBFGSX_GRAM=i8) reads the compact copy of the free rows and writes it when its pass
    rebuilds it (it used to leave the sweeps on the masked full-length columns).  At m = 15 -- where the carried first
    solve does not apply and EVERY iteration runs a full pass, the tile kernel in its largest class -- a run with it, one
    with it from 23 columns on only, and the double-double default must be the same run, and the oracle's to 1e-10."""
    n, m, iters = 200000, 15, 24      # > 4096 free rows: the compact copy is in use; 2c reaches 30 columns
    a, b = O.quad_problem(n, 10.0, 3, O.F64)
    lb, ub = -np.ones(n), np.ones(n)
    res = {}
    for mode in ("default", "i8", "i8-from-23"):
        monkeypatch.delenv("LBFGSX_GRAM", raising=False)
        monkeypatch.delenv("LBFGSX_GRAM_I8_MIN", raising=False)
        if mode != "default":
            monkeypatch.setenv("LBFGSX_GRAM", "i8")
        if mode == "i8-from-23":
            monkeypatch.setenv("LBFGSX_GRAM_I8_MIN", "23")
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters))
        x = np.zeros(n)
        tr = A.TraceBuffer(n, cap=512, stride=max(1, n // 5000))
        niter, fx = s.minimize(A.DiagQuadratic(a, b), x, lb, ub, trace=tr)
        res[mode] = (niter, s.last.nfev, fx, x, tr.xs[:tr.count].copy())
        s.close()
    for mode in ("i8", "i8-from-23"):
        assert res["default"][:3] == res[mode][:3]
        assert np.array_equal(res["default"][3], res[mode][3]) and np.array_equal(res["default"][4], res[mode][4])
    x_ref, r = oracle.lbfgsb(O.F64, O.OBJ_QUAD, np.zeros(n), lb, ub,
                             O.lbfgsb_params(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters), a=a, b=b)
    assert (r.niter, r.nfev) == res["default"][:2] and np.abs(res["default"][3] - x_ref).max() <= 1e-10


@pytest.mark.parametrize("n,m,npairs", [(50000, 10, 10), (300001, 6, 4), (65536, 10, 7), (50000, 12, 12), (70001, 20, 20),
                                        (40000, 40, 40), (30000, 16, 9)])
def test_selected_entries_and_list_grams_equal_the_full_pass(A, n, m, npairs):
    """The pieces of the carried first solve against the full one-pass Gram on the same data: the entries
    lbfgsx_b_gram_pairs_dd returns (one per lane) round to the full pass's entries; with the free set changed by some
    rows, the old un-rounded sums plus the list Gram of the rows that entered / minus that of the rows that left
    (lbfgsx_b_free_delta, lbfgsx_b_gram_list_dd) round to the new full pass's entries.  Misuse is refused."""
    import math

    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    vp, i32, i64, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_double
    ST_FREE, VS_DRT = 1, 0   # LBFGSX_ST_FREE, LBFGSX_VS_DRT (include/lbfgsx.h)
    fdd = core.lbfgsx_b_gram_fused_dd
    fdd.restype, fdd.argtypes = i32, [vp, i32, i32, i32, vp, vp, vp, vp, vp]
    fpairs = core.lbfgsx_b_gram_pairs_dd
    fpairs.restype, fpairs.argtypes = i32, [vp, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp]
    fdelta = core.lbfgsx_b_free_delta
    fdelta.restype, fdelta.argtypes = i32, [vp, C.POINTER(i64), C.POINTER(i64)]
    flist = core.lbfgsx_b_gram_list_dd
    flist.restype, flist.argtypes = i32, [vp, i32, vp]
    fbuild = core.lbfgsx_b_cauchy_build
    fbuild.restype, fbuild.argtypes = i32, [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(f64), vp]
    ffin = core.lbfgsx_b_cauchy_finish
    ffin.restype, ffin.argtypes = i32, [vp, f64, f64, i32, C.POINTER(i64), C.POINTER(i64)]
    h = C.c_void_p()
    L.check(core.lbfgsx_create(C.byref(h), 0, n, m, 0, 1))
    try:
        rng = np.random.default_rng(n + 3)
        for k in range(npairs):
            s_ = rng.standard_normal(n)
            y_ = s_ * (1 + rng.random(n))
            L.check(core.lbfgsx_bfgs_add_correction_host(h, s_.ctypes.data_as(vp), y_.ctypes.data_as(vp)))
        c = min(npairs, m)
        t = 2 * c
        # break points 1 / |g_i| (x0 = 0 in [-1, 1]): lbfgsx_b_cauchy_finish(tc) frees the rows whose break point lies beyond tc
        x0, g = np.zeros(n), rng.standard_normal(n)
        for which, v in ((L.VEC_X, x0), (L.VEC_G, g), (L.VEC_LB, -np.ones(n)), (L.VEC_UB, np.ones(n))):
            L.check(core.lbfgsx_upload(h, which, v.ctypes.data_as(vp)))
        nf, no, dd = i64(), i64(), f64()
        wtd = np.zeros(2 * m)
        L.check(fbuild(h, C.byref(nf), C.byref(no), C.byref(dd), wtd.ctypes.data_as(vp)))

        def free_set(tc):
            na, nfree = i64(), i64()
            L.check(ffin(h, tc, tc, 0, C.byref(na), C.byref(nfree)))
            assert nfree.value == int((1.0 / np.abs(g) > tc).sum())
            L.check(core.lbfgsx_b_sub_begin(h))            # drt = xcp - x0: the vector the v row is taken with
            return nfree.value

        def full():
            G, w, gd = np.zeros((t, t)), np.zeros(t), np.zeros(t * (t + 1))
            L.check(fdd(h, ST_FREE, VS_DRT, 0, None, None, G.ctypes.data_as(vp), w.ctypes.data_as(vp), gd.ctypes.data_as(vp)))
            return G, w, gd.reshape(-1, 2)

        n1 = free_set(0.8)
        ne, nl = i64(), i64()
        L.check(fdelta(h, C.byref(ne), C.byref(nl)))         # first call: everything "entered"
        assert nl.value == 0 and ne.value in (n1, -1)
        G1, w1, gd1 = full()
        # the rows of one slot's two columns and the v row, one entry per lane, as carried_gram asks for them
        ds = c // 2
        pi, pj = [], []
        for J in range(t):
            pi.append(max(ds, J)); pj.append(min(ds, J))
        for J in range(t):
            if J != ds:
                pi.append(max(c + ds, J)); pj.append(min(c + ds, J))
        vrow = len(pi)
        for J in range(t + 1):
            pi.append(t); pj.append(J)
        core.lbfgsx_b_gram_pairs_max.restype, core.lbfgsx_b_gram_pairs_max.argtypes = i32, [vp]
        assert core.lbfgsx_b_gram_pairs_max(h) == 3 * (t + 1)     # the split-row kernel: the v row and the rows of two columns
        if len(pi) <= core.lbfgsx_b_gram_pairs_max(h):
            api, apj = (i32 * len(pi))(*pi), (i32 * len(pi))(*pj)
            pd = np.zeros(2 * len(pi))
            L.check(fpairs(h, ST_FREE, VS_DRT, 0, None, None, len(pi), api, apj, -2, pd.ctypes.data_as(vp)))
            got = pd[0::2] + pd[1::2]
            for e in range(vrow):
                assert got[e] == G1[pi[e], pj[e]], (e, pi[e], pj[e])
            assert np.array_equal(got[vrow:vrow + t], w1)
            assert fpairs(h, ST_FREE, VS_DRT, 0, None, None, 0, api, apj, -2, pd.ctypes.data_as(vp)) == L.E_INVALID
            assert fpairs(h, ST_FREE, VS_DRT, 0, None, None, 3, api, apj, 99, pd.ctypes.data_as(vp)) == L.E_INVALID
        tri = [(i, j) for i in range(t) for j in range(i + 1)]
        for tc, grows in ((0.79, True), (0.805, False)):     # a few more rows free, then fewer (the lists hold max(2^14, n/64) rows)
            nprev = free_set(tc)
            L.check(fdelta(h, C.byref(ne), C.byref(nl)))
            assert (ne.value > 0 and nl.value == 0) if grows else (ne.value == 0 and nl.value > 0)
            ldd = np.zeros(t * (t + 1))
            L.check(flist(h, 0 if grows else 1, ldd.ctypes.data_as(vp)))
            assert flist(h, 1 if grows else 0, ldd.ctypes.data_as(vp)) == L.E_INVALID   # the other list is empty
            ldd = ldd.reshape(-1, 2)
            G2, w2, gd2 = full()
            sign = 1.0 if grows else -1.0
            for e, (i, j) in enumerate(tri):
                want = math.fsum([gd1[e, 0], gd1[e, 1], sign * ldd[e, 0], sign * ldd[e, 1]])
                assert want == G2[i, j], "entry (%d, %d): %.17g != %.17g" % (i, j, want, G2[i, j])
            gd1 = gd2
        assert flist(h, 2, ldd.ctypes.data_as(vp)) == L.E_INVALID
        flu = core.lbfgsx_b_lu_sweep
        flu.restype, flu.argtypes = i32, [vp, vp, f64, vp]
        s7 = (i64 * 7)()
        assert flu(h, None, 1.0, C.cast(s7, vp)) == L.E_INVALID          # nothing to complete
        fss = core.lbfgsx_b_solve_sweep
        fss.restype, fss.argtypes = i32, [vp, i32, i32, vp, f64, vp, vp]
        assert fss(h, 0, VS_DRT, None, 1.0, wtd.ctypes.data_as(vp), C.cast(s7, vp)) == L.E_INVALID   # no index list yet
    finally:
        core.lbfgsx_destroy(h)


@pytest.mark.rhs_pass
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n,m,iters,max_submin", [(70001, 3, 30, 10), (70001, 8, 40, 10), (90000, 10, 45, 10), (70001, 12, 40, 10),
                                                  (65536, 16, 50, 10), (90000, 20, 60, 10), (65536, 40, 100, 10), (70001, 10, 40, 2)])
def test_rows_split_over_lanes_change_no_bit(A, monkeypatch, n, m, iters, max_submin, dtype):
    """The passes of the subspace minimisation and of the Cauchy dots with a row's 2c columns split over 2 or 4 lanes
    (csrc/lbfgsb_x.cuh, every 2c <= 80) against the round-3 kernels that give a lane a whole row (LBFGSX_SPLIT=0; beyond
    2c = 20 / 24 / 32 those hand over to the multi-launch forms): the same statements -- the left-to-right sums of the prologue
    and of the solve handed from group to group in column order -- and the same correctly rounded dots, so the same
    trajectory bit for bit, the same sweeps; the carried Gram must have run at every m."""
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    a, b = O.quad_problem(n, 30.0, 13, dt)
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("LBFGSX_SPLIT", on)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters, max_submin=max_submin),
                           dtype=npdt)
        tr = A.TraceBuffer(n, cap=512, stride=23)
        x = np.zeros(n, dtype=npdt)
        try:
            niter, fx = s.minimize(A.DiagQuadratic(a, b), x, (-0.7 * np.ones(n)).astype(npdt), (0.9 * np.ones(n)).astype(npdt),
                                   trace=tr)
        except RuntimeError:
            niter, fx = -1, float("nan")
        st = s.stats()
        res[on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), st["submin_sweeps"], st["submin_unconverged"],
                   st["gram_carried"], st["submin_calls"])
    f, u = res["1"], res["0"]
    assert f[:2] == u[:2] and f[4:6] == u[4:6] and f[4] > 0
    assert np.array_equal(f[2], u[2]) and np.array_equal(f[3], u[3])
    if dtype == "f64":
        assert f[6] >= f[7] // 4, "the carried form ran in %d of %d subspace minimisations" % (f[6], f[7])
    if m > 10:
        assert u[6] <= 10   # the one-entry-per-lane kernel has 64 lanes: only while the history holds <= 10 pairs


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n,m,iters,cap", [(70001, 8, 40, None), (90000, 10, 45, None), (65536, 20, 50, None), (120000, 10, 40, "8")])
def test_cauchy_finish_carrying_the_next_statements_changes_no_bit(A, monkeypatch, n, m, iters, cap, dtype):
    """lbfgsx_b_cauchy_finish also evaluates drt = xcp - x0 (the statement that opens the subspace minimisation,
    SubspaceMin.h:130) on the values it holds, and lists the rows it made newly active so that W_A'(A'd) of compute_FtBAb
    (BFGSMat.h:503-507) walks a list instead of n state bytes -- against the separate passes (LBFGSX_FINISH_FUSE=0): the same
    statements, the same correctly rounded dots, so the same trajectory bit for bit.  cap = 8: the list overflows and the scan
    takes over."""
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    a, b = O.quad_problem(n, 30.0, 17, dt)
    if cap:
        monkeypatch.setenv("LBFGSX_NEWACT_CAP", cap)
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("LBFGSX_FINISH_FUSE", on)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters), dtype=npdt)
        tr = A.TraceBuffer(n, cap=512, stride=19)
        x = np.zeros(n, dtype=npdt)
        try:
            niter, fx = s.minimize(A.DiagQuadratic(a, b), x, (-0.7 * np.ones(n)).astype(npdt), (0.9 * np.ones(n)).astype(npdt),
                                   trace=tr)
        except RuntimeError:
            niter, fx = -1, float("nan")
        st = s.stats()
        res[on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), st["submin_sweeps"], st["gcp_crossings"])
    f, u = res["1"], res["0"]
    assert f[:2] == u[:2] and f[4:] == u[4:]
    assert np.array_equal(f[2], u[2]) and np.array_equal(f[3], u[3])


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n,m,iters", [(70001, 8, 40), (300000, 10, 45), (90000, 20, 50), (65536, 40, 70)])
def test_sums_over_the_L_u_U_list_in_one_pass_change_no_bit(A, monkeypatch, n, m, iters, dtype):
    """A BOXCQP sweep's W_{L u U}'(-c) as a third set of sums inside the pass that computes W_L'l and W_U'u over the same index
    list (kx_list2<..., WITHC>) against the launch of its own that preceded that pass (LBFGSX_LIST12=0: kx_list1): the same rows
    in the same lanes, so the same un-rounded (hi, lo) pairs -- which the "W_P'rhs without a pass" identity subtracts from
    others before rounding (BFGSMatB::solve_PtBP) -- and the same trajectory bit for bit; the identity must have run."""
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    a, b = O.quad_problem(n, 30.0, 41, dt)
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("LBFGSX_LIST12", on)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters), dtype=npdt)
        tr = A.TraceBuffer(n, cap=512, stride=37)
        x = np.zeros(n, dtype=npdt)
        try:
            niter, fx = s.minimize(A.DiagQuadratic(a, b), x, (-0.7 * np.ones(n)).astype(npdt), (0.9 * np.ones(n)).astype(npdt),
                                   trace=tr)
        except RuntimeError:
            niter, fx = -1, float("nan")
        st = s.stats()
        res[on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), st["submin_sweeps"], st["gcp_crossings"],
                   st["rhs_identities"])
    f, u = res["1"], res["0"]
    assert f[:2] == u[:2] and f[4:] == u[4:] and f[0] > 0
    assert np.array_equal(f[2], u[2]) and np.array_equal(f[3], u[3])
    if dtype == "f64":
        assert f[6] > 0, "no sweep took the identity that reads these sums"


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n,m,iters,ties", [(70001, 8, 40, False), (300000, 10, 45, False), (120000, 6, 40, True)])
def test_short_candidate_lists_ordered_by_one_block_change_no_bit(A, monkeypatch, n, m, iters, ties, dtype):
    """The partial sort of <= 4096 listed candidates in one block (k_psel_sort_small: (key, row) pairs ordered in LDS by the
    radix sort's key order, rows ascending among equal keys) against the three launches it replaces (LBFGSX_PSEL_SMALL=0: radix
    sort of the rows, gather of the keys, stable radix sort by key): the same sorted break points, so the same searches and the
    same trajectory bit for bit (lbfgsx_b_psel_counts tells that the one-block form ran).  ties: blocks of coordinates share a,
    b, x0 and the bounds, so whole groups of break points are EQUAL and only the row order separates them (Cauchy.h:193-199
    walks them in that order)."""
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    a, b = O.quad_problem(n, 30.0, 29, dt)
    if ties:
        a = np.repeat(a[::8], 8)[:n].copy()
        b = np.repeat(b[::8], 8)[:n].copy()
    from lbfgspp_amd import _lib as L
    core, _ = L.load()
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("LBFGSX_PSEL_SMALL", on)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters), dtype=npdt)
        tr = A.TraceBuffer(n, cap=512, stride=31)
        x = np.zeros(n, dtype=npdt)
        pc = (C.c_longlong * 1)()
        core.lbfgsx_b_psel_counts(None, 1)
        try:
            niter, fx = s.minimize(A.DiagQuadratic(a, b), x, (-0.7 * np.ones(n)).astype(npdt), (0.9 * np.ones(n)).astype(npdt),
                                   trace=tr)
        except RuntimeError:
            niter, fx = -1, float("nan")
        core.lbfgsx_b_psel_counts(C.byref(pc), 0)
        st = s.stats()
        res[on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), st["submin_sweeps"], st["gcp_crossings"],
                   st["gcp_partial_sorts"], list(pc))
    f, u = res["1"], res["0"]
    assert f[:2] == u[:2] and f[4:7] == u[4:7] and f[0] > 0
    assert np.array_equal(f[2], u[2]) and np.array_equal(f[3], u[3])
    assert f[6] > 0, "no search took the partial sort"
    assert f[7][0] > 0 and u[7] == [0], "candidate lists ordered by one block: %s with, %s without" % (f[7], u[7])


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n,m,iters,env", [(70001, 8, 40, {}), (90000, 10, 45, {}), (65536, 20, 50, {}),
                                           (120000, 10, 40, {"LBFGSX_GCP_TAU_FACTOR": "0"}),
                                           (120000, 6, 30, {"LBFGSX_SELECT_INLINE": "0"}),
                                           (300000, 10, 30, {"LBFGSX_FORCE_FUSE": "0"})])
def test_post_statements_carrying_the_cauchy_build_change_no_bit(A, monkeypatch, n, m, iters, env, dtype):
    """lbfgsx_b_post_linesearch_build: the statements after the line search (LBFGSB.h:206,235-237) and the element-wise part
    of the Cauchy search of the same iteration (Cauchy.h:95,111-129) in one pass over x and g -- against the two passes of
    rounds 1-3 (LBFGSX_POST_BUILD=0): the same statements on the same operands, so the same trajectory bit for bit, and the
    searches did use what the post pass left them.  Without a threshold for the partial sort (TAU_FACTOR=0) the full sort
    follows; with the selection behind the build (SELECT_INLINE=0) and with the clamp as a pass of its own (FORCE_FUSE=0:
    x may have moved) the two passes stay, by themselves."""
    import ctypes as C
    core, _ = A.load()
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    a, b = O.quad_problem(n, 30.0, 17, dt)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("LBFGSX_POST_BUILD", on)
        core.lbfgsx_b_post_build_counts(None, 1)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters), dtype=npdt)
        tr = A.TraceBuffer(n, cap=512, stride=19)
        x = np.zeros(n, dtype=npdt)
        try:
            niter, fx = s.minimize(A.DiagQuadratic(a, b), x, (-0.7 * np.ones(n)).astype(npdt), (0.9 * np.ones(n)).astype(npdt),
                                   trace=tr)
        except RuntimeError:
            niter, fx = -1, float("nan")
        st = s.stats()
        pc = (C.c_int64 * 2)()
        core.lbfgsx_b_post_build_counts(pc, 0)
        res[on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), st["submin_sweeps"], st["gcp_crossings"], st["gcp_sorted"],
                   tuple(pc))
    f, u = res["1"], res["0"]
    assert f[:2] == u[:2] and f[4:7] == u[4:7]
    assert np.array_equal(f[2], u[2]) and np.array_equal(f[3], u[3])
    assert u[7] == (0, 0)
    if "LBFGSX_FORCE_FUSE" in env:
        assert f[7][1] == 0 or f[7][1] <= f[7][0]       # an explicit clamp pass in between drops what the post pass prepared
    elif "LBFGSX_SELECT_INLINE" in env:
        assert f[7][1] <= f[7][0]
    elif f[0] > 3:
        assert f[7][0] >= f[0] - 1 and f[7][1] >= f[0] - 3, f[7]   # every iteration but the last searches with it


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("obj,n,m,iters", [("quad", 70001, 8, 40), ("quad", 90000, 10, 45), ("rosen", 65536, 6, 40),
                                           ("rosen", 100002, 20, 50)])
def test_first_trial_evaluated_with_dg_and_step_max_changes_no_bit(A, monkeypatch, obj, n, m, iters, dtype):
    """lbfgsx_b_dg_maxstep_trial: the pass that works out dg and step_max (LBFGSB.h:176-179) also evaluates the line search's
    first trial at min(1, max_step) (:200-203, LineSearchMoreThuente.h:261-262) and lbfgsx_trial hands it over when the
    search asks for exactly that step -- against the two passes (LBFGSX_TRIAL_AHEAD=0): the same statements on the same
    operands, the same trajectory bit for bit; most searches do start at step 1 and use it, the others (step_max < 1) throw
    it away and the next four iterations go without.  n = 70001: a scalar tail."""
    import ctypes as C
    core, _ = A.load()
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    if obj == "quad":
        a, b = O.quad_problem(n, 30.0, 17, dt)
        f = A.DiagQuadratic(a, b)
        x0 = np.zeros(n, dtype=npdt)
        lb, ub = (-0.7 * np.ones(n)).astype(npdt), (0.9 * np.ones(n)).astype(npdt)
    else:
        f = A.ExtendedRosenbrock()
        x0 = O.rosen_x0(n, 5, dt)
        lb, ub = (-1.5 * np.ones(n)).astype(npdt), (0.8 * np.ones(n)).astype(npdt)
        x0 = np.minimum(np.maximum(x0, lb), ub)
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("LBFGSX_TRIAL_AHEAD", on)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters), dtype=npdt)
        tr = A.TraceBuffer(n, cap=512, stride=19)
        x = x0.copy()
        try:
            niter, fx = s.minimize(f, x, lb, ub, trace=tr)
        except RuntimeError:
            niter, fx = -1, float("nan")
        pc = (C.c_int64 * 2)()
        core.lbfgsx_b_trial_ahead_counts(s.ctx, pc)
        res[on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), tr.fx[:tr.count].copy(), tuple(pc))
    f1, f0 = res["1"], res["0"]
    assert f1[:2] == f0[:2]
    assert np.array_equal(f1[2], f0[2]) and np.array_equal(f1[3], f0[3]) and np.array_equal(f1[4], f0[4])
    assert f0[5] == (0, 0)
    if f1[0] > 8:
        assert f1[5][0] >= f1[0] // 3 and f1[5][1] >= f1[5][0] // 2, f1[5]


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n,m,iters", [(70001, 8, 40), (90000, 10, 45), (65536, 20, 50), (400000, 10, 30)])
def test_sweep_rhs_products_from_held_sums_stay_within_an_ulp_of_the_pass(A, boracle, monkeypatch, n, m, iters, dtype):
    """A BOXCQP sweep's solve opens with W_P' rhs (BFGSMat.h:560) after rhs_P = c_P + B[P,L] l + B[P,U] u (SubspaceMin.h:232-241).
    Rounds 1-3 made a pass over P for those 2c numbers; now they come from sums the host holds un-rounded -- W_F'(-c) of the
    first solve, W_{L u U}'(-c) and the Gram of the complement identity -- and the row-wise rhs is written by the solve's own
    pass (lbfgsx_b_solve_sweep_rhs).  NOT bit-identical: the pass summed the rounded rows, this is the exact sum rounded once,
    a difference in the last bit of some of the 2c numbers, which the 2c x 2c solve carries into y.  So: the same iteration and
    evaluation counts and the same sweeps as LBFGSX_RHS_IDENTITY=0, iterates within 1e-10 (f64; north_star's tolerance) of it
    at every evaluation -- and both as close to the reference (extended-precision sums) as each other."""
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    a, b = O.quad_problem(n, 30.0, 17, dt)
    lb, ub = (-0.7 * np.ones(n)).astype(npdt), (0.9 * np.ones(n)).astype(npdt)
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("LBFGSX_RHS_IDENTITY", on)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters), dtype=npdt)
        tr = A.TraceBuffer(n, cap=512, stride=19)
        x = np.zeros(n, dtype=npdt)
        niter, fx = s.minimize(A.DiagQuadratic(a, b), x, lb, ub, trace=tr)
        st = s.stats()
        res[on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), st["submin_sweeps"], st["rhs_identities"])
    f, u = res["1"], res["0"]
    tol = 1e-10 if dtype == "f64" else 2e-4
    assert u[5] == 0
    if dtype == "f64":
        assert f[:2] == u[:2] and f[4] == u[4]
        assert f[5] > 0 and f[5] >= f[4] // 4, (f[4], f[5])      # most sweeps of the iterations with a carried Gram
        assert np.abs(f[3] - u[3]).max() <= tol and np.abs(f[2] - u[2]).max() <= tol
        p = O.lbfgsb_params(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters)
        x_ref, r_ref = boracle.lbfgsb(dt, O.OBJ_QUAD, np.zeros(n, dtype=npdt), lb, ub, p, a=a, b=b)
        assert (r_ref.niter, r_ref.nfev) == f[:2]
        d1, d0 = np.abs(f[2] - x_ref).max(), np.abs(u[2] - x_ref).max()
        assert d1 <= tol and d0 <= tol, (d1, d0)
    else:
        # f32: a last-bit difference in W_P' rhs may move a line search by an evaluation; the minimiser stays the same
        assert f[5] > 0
        assert np.abs(f[2] - u[2]).max() <= tol


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n,m,iters,lo,hi", [(70001, 8, 40, -0.7, 0.9), (90000, 20, 60, -0.7, 0.9), (65536, 10, 45, -0.05, 30.0),
                                            (120000, 16, 50, -30.0, 0.05)])
def test_list_pass_over_L_u_U_with_one_empty_set_changes_no_bit(A, monkeypatch, n, m, iters, lo, hi, dtype):
    """A BOXCQP sweep needs W_L' l and W_U' u (SubspaceMin.h:236-241).  The pass over the index list of L u U delivers both and
    leaves the sums the solve's identity uses; until round 6 it ran only when BOTH sets were non-empty, and late sweeps -- a
    handful of rows on one side -- fell back to a masked pass over all rows plus a Gram pass over P (m = 20: 0.6 ms of a 3 ms
    iteration).  Now it runs whenever the list is non-empty (LBFGSX_LU_ONE_SIDED=0: as before); an empty set contributes a zero
    sum and a zero count, which is what the per-set form answers for it.  With the identity off (LBFGSX_RHS_IDENTITY=0) the two
    forms must agree bit for bit -- same dots by order-independent sums; with it on, to the identity's own tolerance
    (test_sweep_rhs_products_from_held_sums...).  The last two boxes are one-sided by construction (only one bound is ever met)."""
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    a, b = O.quad_problem(n, 30.0, 17, dt)
    lb, ub = (lo * np.ones(n)).astype(npdt), (hi * np.ones(n)).astype(npdt)
    res = {}
    for ident in ("0", "1"):
        for on in ("1", "0"):
            monkeypatch.setenv("LBFGSX_RHS_IDENTITY", ident)
            monkeypatch.setenv("LBFGSX_LU_ONE_SIDED", on)
            s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters), dtype=npdt)
            tr = A.TraceBuffer(n, cap=512, stride=19)
            x = np.zeros(n, dtype=npdt)
            niter, fx = s.minimize(A.DiagQuadratic(a, b), x, lb, ub, trace=tr)
            st = s.stats()
            res[ident, on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), st["submin_sweeps"], st["rhs_identities"])
    f, u = res["0", "1"], res["0", "0"]
    assert f[:2] == u[:2] and f[4] == u[4] and f[4] > 0
    assert np.array_equal(f[2], u[2]) and np.array_equal(f[3], u[3])
    g, v = res["1", "1"], res["1", "0"]
    assert g[5] >= v[5]                     # the identity serves at least the sweeps it served before
    tol = 1e-10 if dtype == "f64" else 2e-4
    assert np.abs(g[2] - f[2]).max() <= tol
    if dtype == "f64":
        assert g[:2] == f[:2] and g[4] == f[4] and np.abs(g[3] - f[3]).max() <= tol


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n,m,iters", [(70001, 8, 40), (300000, 10, 30), (65536, 20, 40)])
def test_first_chunk_of_the_break_points_gathered_ahead_changes_no_bit(A, monkeypatch, n, m, iters, dtype):
    """The host form of the Cauchy search opens with the first 512 sorted break points (Cauchy<Scalar>::Stream).  Their gather
    and copy ride behind the build's sort, ahead of its W'd pass, and have landed when that pass's wait returns
    (LBFGSX_CHUNK_AHEAD=0: a round trip of their own, as before).  Same data either way: same trajectory bit for bit, one
    wait less per iteration."""
    import ctypes as C
    core, _ = A.load()
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    a, b = O.quad_problem(n, 30.0, 17, dt)
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("LBFGSX_CHUNK_AHEAD", on)
        cnt0 = (C.c_int64 * 3)()
        core.lbfgsx_counters(cnt0, 1)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters), dtype=npdt)
        tr = A.TraceBuffer(n, cap=512, stride=19)
        x = np.zeros(n, dtype=npdt)
        niter, fx = s.minimize(A.DiagQuadratic(a, b), x, (-0.7 * np.ones(n)).astype(npdt), (0.9 * np.ones(n)).astype(npdt), trace=tr)
        cnt = (C.c_int64 * 3)()
        core.lbfgsx_counters(cnt, 0)
        st = s.stats()
        res[on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), st["gcp_crossings"], int(cnt[1]))
    f, u = res["1"], res["0"]
    assert f[:2] == u[:2] and f[4] == u[4]
    assert np.array_equal(f[2], u[2]) and np.array_equal(f[3], u[3])
    assert f[5] <= u[5] - (f[0] - m - 2), (f[5], u[5])   # a stream wait less in (at least) every iteration with a full history


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("n,m,iters", [(70001, 8, 40), (300000, 10, 30), (65536, 20, 40)])
def test_free_set_delta_launched_ahead_changes_no_bit(A, monkeypatch, n, m, iters, dtype):
    """lbfgsx_b_free_delta (the rows that entered / left the free set since the last iteration: what the carried Gram patches
    its sums with) needs nothing from the host, so it rides ahead of the pass over the newly active rows of the same
    subspace minimisation and its counters are there when that pass's wait returns (LBFGSX_DELTA_AHEAD=0: launched on
    request, with a stream wait of its own).  Same kernels on the same state bytes: same trajectory bit for bit, fewer
    stream waits, and the carried form runs as often."""
    import ctypes as C
    core, _ = A.load()
    dt = O.F64 if dtype == "f64" else O.F32
    npdt = O.NPDT[dt]
    a, b = O.quad_problem(n, 30.0, 17, dt)
    res = {}
    for on in ("1", "0"):
        monkeypatch.setenv("LBFGSX_DELTA_AHEAD", on)
        cnt0 = (C.c_int64 * 3)()
        core.lbfgsx_counters(cnt0, 1)
        s = A.LBFGSBSolver(A.LBFGSBParam(m=m, epsilon=0, epsilon_rel=0, past=0, max_iterations=iters), dtype=npdt)
        tr = A.TraceBuffer(n, cap=512, stride=19)
        x = np.zeros(n, dtype=npdt)
        niter, fx = s.minimize(A.DiagQuadratic(a, b), x, (-0.7 * np.ones(n)).astype(npdt), (0.9 * np.ones(n)).astype(npdt), trace=tr)
        cnt = (C.c_int64 * 3)()
        core.lbfgsx_counters(cnt, 0)
        st = s.stats()
        res[on] = (niter, s.last.nfev, x.copy(), tr.xs[:tr.count].copy(), st["gram_carried"], st["submin_sweeps"], int(cnt[1]))
    f, u = res["1"], res["0"]
    assert f[:2] == u[:2] and f[4:6] == u[4:6]
    assert np.array_equal(f[2], u[2]) and np.array_equal(f[3], u[3])
    assert f[6] < u[6], (f[6], u[6])
