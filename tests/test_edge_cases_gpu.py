"""-m gpu: edge cases the reference's examples and parameter space imply -- tiny / odd dimensions, m larger than
the iteration count, infinite and degenerate bounds, f32 L-BFGS-B, exhausted line searches, early exits."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import lbfgspp_amd as A
    A.load()
    return A


def _run_lbfgs(A, dtype, ls, obj, x0, a=None, b=None, **pk):
    s = A.LBFGSSolver(A.LBFGSParam(**pk), linesearch=ls, dtype=O.NPDT[dtype])
    x = np.array(x0, dtype=O.NPDT[dtype])
    f = A.DiagQuadratic(a, b) if obj == O.OBJ_QUAD else A.ExtendedRosenbrock()
    try:
        niter, fx = s.minimize(f, x)
        return x, niter, s.last.nfev, 0, fx, ""
    except (RuntimeError, ArithmeticError, ValueError) as e:
        return x, s.last.niter, s.last.nfev, s.last.status, s.last.fx, str(e)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 63, 64, 65, 255, 257])
@pytest.mark.parametrize("dtype", [O.F64, O.F32])
def test_small_and_ragged_dimensions_quadratic(A, oracle, n, dtype):
    a, b = O.quad_problem(n, 7.0, 2, dtype)
    x0 = np.linspace(-1, 1, n).astype(O.NPDT[dtype])
    p = O.lbfgs_params(m=4, max_iterations=30)
    x_ref, r = oracle.lbfgs(dtype, O.LS_NW, O.OBJ_QUAD, x0, p, a=a, b=b)
    x, niter, nfev, status, fx, msg = _run_lbfgs(A, dtype, O.LS_NW, O.OBJ_QUAD, x0, a, b, m=4, max_iterations=30)
    assert (status != 0) == (r.status != 0)
    if r.status == 0:
        assert (niter, nfev) == (r.niter, r.nfev)
        assert np.array_equal(x, x_ref)
    else:
        assert msg == r.msg.decode()


def test_rosenbrock_requires_even_dimension(A):
    s = A.LBFGSSolver(A.LBFGSParam())
    with pytest.raises(ValueError, match="even dimension"):
        s.minimize(A.ExtendedRosenbrock(), np.zeros(7))


def test_history_longer_than_run(A, oracle):
    n = 500
    x0 = O.rosen_x0(n)
    p = O.lbfgs_params(m=25, epsilon=0, epsilon_rel=0, max_iterations=8)
    x_ref, r = oracle.lbfgs(O.F64, O.LS_MT, O.OBJ_ROSEN, x0, p)
    x, niter, nfev, status, fx, _ = _run_lbfgs(A, O.F64, O.LS_MT, O.OBJ_ROSEN, x0, m=25, epsilon=0, epsilon_rel=0,
                                              max_iterations=8)
    assert status == 0 and (niter, nfev) == (r.niter, r.nfev) and np.array_equal(x, x_ref)


def test_m_equal_one(A, oracle):
    n = 2000
    a, b = O.quad_problem(n)
    p = O.lbfgs_params(m=1, epsilon=0, epsilon_rel=0, max_iterations=25)
    x_ref, r = oracle.lbfgs(O.F64, O.LS_NW, O.OBJ_QUAD, np.zeros(n), p, a=a, b=b)
    x, niter, nfev, status, fx, _ = _run_lbfgs(A, O.F64, O.LS_NW, O.OBJ_QUAD, np.zeros(n), a, b, m=1, epsilon=0,
                                              epsilon_rel=0, max_iterations=25)
    assert (niter, nfev) == (r.niter, r.nfev) and np.array_equal(x, x_ref)


def test_start_at_the_minimiser_returns_one_iteration(A):
    """early exit, reference LBFGS.h:100-103"""
    s = A.LBFGSSolver(A.LBFGSParam())
    x = np.ones(10)
    niter, fx = s.minimize(A.ExtendedRosenbrock(), x)
    assert niter == 1 and fx == 0.0 and s.last.nfev == 1 and np.array_equal(x, np.ones(10))


@pytest.mark.parametrize("ls", [O.LS_NW, O.LS_MT, O.LS_BT, O.LS_BR])
def test_exhausted_line_search_behaves_like_reference(A, oracle, ls):
    """max_linesearch = 1..2: MoreThuente / NocedalWright hand back the best point, Backtracking / Bracketing throw."""
    n = 200
    x0 = O.rosen_x0(n)
    for mls in (1, 2):
        p = O.lbfgs_params(m=5, max_iterations=10, max_linesearch=mls)
        x_ref, r = oracle.lbfgs(O.F64, ls, O.OBJ_ROSEN, x0, p)
        x, niter, nfev, status, fx, msg = _run_lbfgs(A, O.F64, ls, O.OBJ_ROSEN, x0, m=5, max_iterations=10,
                                                    max_linesearch=mls)
        assert (status != 0) == (r.status != 0), (ls, mls, msg, r.msg)
        assert nfev == r.nfev
        if r.status == 0:
            assert niter == r.niter and np.array_equal(x, x_ref)
        else:
            assert msg == r.msg.decode()


def test_past_delta_stopping_rule(A, oracle):
    n = 3000
    a, b = O.quad_problem(n)
    for past, delta in ((1, 1e-4), (3, 1e-6)):
        p = O.lbfgs_params(m=6, epsilon=0, epsilon_rel=0, past=past, delta=delta, max_iterations=200)
        x_ref, r = oracle.lbfgs(O.F64, O.LS_NW, O.OBJ_QUAD, np.zeros(n), p, a=a, b=b)
        x, niter, nfev, status, fx, _ = _run_lbfgs(A, O.F64, O.LS_NW, O.OBJ_QUAD, np.zeros(n), a, b, m=6, epsilon=0,
                                                  epsilon_rel=0, past=past, delta=delta, max_iterations=200)
        assert r.niter < 200 and (niter, nfev) == (r.niter, r.nfev) and np.array_equal(x, x_ref)


# ------------------------------------------------------------------ L-BFGS-B
def _run_lbfgsb(A, dtype, obj, x0, lb, ub, a=None, b=None, **pk):
    s = A.LBFGSBSolver(A.LBFGSBParam(**pk), dtype=O.NPDT[dtype])
    x = np.array(x0, dtype=O.NPDT[dtype])
    f = A.DiagQuadratic(a, b) if obj == O.OBJ_QUAD else A.ExtendedRosenbrock()
    niter, fx = s.minimize(f, x, lb, ub)
    return x, niter, s.last.nfev, fx, s


def test_lbfgsb_mixed_infinite_bounds_rosenbrock(A, oracle):
    """bounds pattern of the reference's example-rosenbrock-box.cpp (some coordinates unbounded, some starting on
    their bound) on the pair-form Rosenbrock objective"""
    if not oracle.supports_lbfgsb:
        pytest.skip("oracle without L-BFGS-B")
    n = 600
    x0 = O.rosen_x0(n, 3)
    lb = np.where(np.arange(n) % 3 == 0, -np.inf, -0.5)
    ub = np.where(np.arange(n) % 5 == 0, np.inf, 0.9)
    x0 = np.clip(x0, lb, ub)
    x0[7], x0[11] = ub[7], lb[11]
    p = O.lbfgsb_params(m=5, max_iterations=25)
    x_ref, r = oracle.lbfgsb(O.F64, O.OBJ_ROSEN, x0, lb, ub, p)
    x, niter, nfev, fx, s = _run_lbfgsb(A, O.F64, O.OBJ_ROSEN, x0, lb, ub, m=5, max_iterations=25)
    assert (niter, nfev) == (r.niter, r.nfev)
    assert np.abs(x - x_ref).max() <= 1e-10
    assert np.all(x >= lb) and np.all(x <= ub)


def test_lbfgsb_fixed_variables_and_tiny_problems(A, oracle):
    if not oracle.supports_lbfgsb:
        pytest.skip("oracle without L-BFGS-B")
    for n in (1, 2, 9, 130):
        a, b = O.quad_problem(n, 5.0, 4)
        lb, ub = -0.3 * np.ones(n), 0.4 * np.ones(n)
        lb[::4] = ub[::4] = 0.1  # fixed coordinates (lb == ub)
        x0 = np.zeros(n)
        p = O.lbfgsb_params(m=3, max_iterations=30)
        x_ref, r = oracle.lbfgsb(O.F64, O.OBJ_QUAD, x0, lb, ub, p, a=a, b=b)
        x, niter, nfev, fx, s = _run_lbfgsb(A, O.F64, O.OBJ_QUAD, x0, lb, ub, a, b, m=3, max_iterations=30)
        assert (niter, nfev) == (r.niter, r.nfev)
        assert np.abs(x - x_ref).max() <= 1e-10 and np.all(x[::4] == 0.1)


def test_lbfgsb_float32(A, oracle):
    if not oracle.supports_lbfgsb:
        pytest.skip("oracle without L-BFGS-B")
    n = 4000
    a, b = O.quad_problem(n, 10.0, 1, O.F32)
    lb, ub = -np.ones(n, np.float32), np.ones(n, np.float32)
    p = O.lbfgsb_params(m=6, epsilon=0, epsilon_rel=0, past=0, max_iterations=8)
    x_ref, r = oracle.lbfgsb(O.F32, O.OBJ_QUAD, np.zeros(n, np.float32), lb, ub, p, a=a, b=b)
    x, niter, nfev, fx, s = _run_lbfgsb(A, O.F32, O.OBJ_QUAD, np.zeros(n, np.float32), lb, ub, a, b, m=6, epsilon=0,
                                        epsilon_rel=0, past=0, max_iterations=8)
    assert niter == r.niter
    assert np.abs(x.astype(np.float64) - x_ref.astype(np.float64)).max() <= 1e-4


def test_lbfgsb_unconstrained_limit_matches_free_solution(A):
    """with bounds far away L-BFGS-B must reach the unconstrained minimiser b/a"""
    n = 5000
    a, b = O.quad_problem(n)
    s = A.LBFGSBSolver(A.LBFGSBParam(m=8, epsilon=1e-8, epsilon_rel=0.0, past=0, max_iterations=300))
    x = np.zeros(n)
    niter, fx = s.minimize(A.DiagQuadratic(a, b), x, -1e6 * np.ones(n), 1e6 * np.ones(n))
    assert niter < 300 and np.abs(x - b / a).max() < 1e-6


def test_context_creation_failure_is_reported_and_leaves_the_device_usable(A):
    """a dimension no HBM can hold: lbfgsx_create must answer LBFGSX_E_HIP (no crash, nothing leaked: a normal
    context of the north-star's per-vector size can still be created right after)"""
    import ctypes as C
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    h = C.c_void_p()
    rc = core.lbfgsx_create(C.byref(h), O.F64, 1 << 40, 10, 0, L.FLAG_BOUNDED)
    assert rc == L.E_HIP and not h.value
    assert b"hipMalloc" in core.lbfgsx_last_error()
    for _ in range(3):  # repeated failures must not accumulate allocations
        assert core.lbfgsx_create(C.byref(h), O.F64, 1 << 40, 4, 0, 0) == L.E_HIP
    L.check(core.lbfgsx_create(C.byref(h), O.F64, 100_000_000, 2, 0, 0))
    core.lbfgsx_destroy(h)
    for bad in ((O.F64, 0, 5), (O.F64, 10, 0), (7, 10, 5)):
        assert core.lbfgsx_create(C.byref(h), bad[0], bad[1], bad[2], 0, 0) == L.E_INVALID


def test_no_device_memory_is_leaked_across_solver_lifetimes(A):
    """every mode allocates its work sets lazily (L-BFGS-B sort / Cauchy scratch, Gram-space scratch, f32 history, batch
    buffers): 20 create / solve / destroy cycles must hand all of it back"""
    import torch
    from lbfgspp_amd import _lib as L
    from lbfgspp_amd import batched as B
    n = 1 << 20

    def cycle():
        s = A.LBFGSSolver(A.LBFGSParam(m=5, epsilon=0.0, epsilon_rel=0.0, max_iterations=6), linesearch=A.LS_MORE_THUENTE)
        for form in (L.RECURSION_VECTOR, L.RECURSION_GRAM_SPACE, L.RECURSION_GRAM_SPACE_F32H):
            s.set_recursion(form)
            s.minimize(A.ExtendedRosenbrock(), O.rosen_x0(n))
        s.close()
        a, b = O.quad_problem(n)
        sb = A.LBFGSBSolver(A.LBFGSBParam(m=5, max_iterations=6))
        sb.minimize(A.DiagQuadratic(a, b), np.zeros(n), -np.ones(n), np.ones(n))
        sb.close()
        B.solve_local_lockstep(A.LBFGSParam(m=4, epsilon=0.0, epsilon_rel=0.0, max_iterations=3), 4096, 0, 8,
                               dtype=np.float32)

    cycle()  # first use loads code objects and sizes the allocator's pools
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(20):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, "device memory shrank by %.1f MB over 20 cycles" % ((free0 - free1) / 2**20)


@pytest.mark.gpu
@pytest.mark.parametrize("f32acc", [0, 1])
@pytest.mark.parametrize("grid", [1, 2, 3, 5, 64, 257, 1000])
def test_grid_reduction_routes_every_partial_to_its_sum(grid, f32acc):
    """csrc/reduce.cuh directly: `nred` sums over `grid` blocks whose terms depend on the sum's index, the thread and the
    block (lbfgsx_selftest_reduce) against the closed form -- the recursive halving across the lanes, the per-sum threads,
    the batched partial loads of the last block and the one-block shortcut, for every count of sums the kernels use and
    the odd ones in between."""
    import ctypes as C

    import lbfgspp_amd as A
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    ctx = C.c_void_p()
    L.check(core.lbfgsx_create(C.byref(ctx), L.F64, 4096, 3, 0, 0))
    try:
        g = np.arange(256 * grid, dtype=np.int64)
        for nred in (1, 2, 3, 5, 7, 8, 9, 25, 31, 33, 40, 50, 56):
            out = (C.c_double * 64)()
            L.check(core.lbfgsx_selftest_reduce(ctx, nred, grid, f32acc, out))
            for r in range(nred):
                want = int(((r + 1) * (1 + g % 7) + ((g // 256) % 3) + (r * 1000003 % 17)).sum())
                assert out[r] == float(want), "nred %d grid %d sum %d: %r != %d" % (nred, grid, r, out[r], want)
        bad = (C.c_double * 64)()
        assert core.lbfgsx_selftest_reduce(ctx, 4, grid, 0, bad) == L.E_INVALID
    finally:
        core.lbfgsx_destroy(ctx)
