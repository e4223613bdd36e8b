"""-m gpu: the Gram-space ("vector-free") form of the two-loop recursion (lbfgspp_amd/csrc/gram_space.hip,
include/LBFGSpp/GramSpace.h).  The mode is opt-in and outside the bit-parity contract, so the checks are: the two kernels
against numpy on identical inputs, the Gram-space direction against the vector two-loop (reference BFGSMat.h:276-302,
the parity-tested lbfgsx_apply_Hv) on the SAME device history, and whole runs against the vector form of the solver."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import lbfgspp_amd as A
    core, _ = A.load()
    assert core.lbfgsx_device_count() >= 1, "no GPU visible: these tests must run on the MI355X box"
    return A


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("dtype", [O.F64, O.F32])
@pytest.mark.parametrize("n,m,npairs", [(4099, 6, 0), (4099, 6, 4), (65537, 10, 10), (1000, 5, 13), (7, 3, 5), (2, 2, 1),
                                        (3001, 20, 23), (513, 24, 24)])
def test_kernels_match_numpy(A, dtype, n, m, npairs):
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    dt = O.NPDT[dtype]
    rng = np.random.default_rng(99 + n + npairs)
    S = rng.standard_normal((max(npairs, 1), n)).astype(dt)[:npairs]
    Y = (rng.standard_normal((max(npairs, 1), n)) * 0.05).astype(dt)[:npairs] + S * dt(1.5)
    h = C.c_void_p()
    L.check(core.lbfgsx_create(C.byref(h), dtype, n, m, 0, 0))
    try:
        for k in range(npairs):
            L.check(core.lbfgsx_bfgs_add_correction_host(h, _vp(S[k]), _vp(Y[k])))
        slot_pair = {k % m: k for k in range(npairs)}  # BFGSMat.h:83: pair k lands in slot k % m
        cn = min(npairs, m)
        xp, gp, d, gt = (rng.standard_normal(n).astype(dt) for _ in range(4))
        step = 0.37
        L.check(core.lbfgsx_upload(h, L.VEC_X, _vp(xp)))
        L.check(core.lbfgsx_upload(h, L.VEC_G, _vp(gp)))
        L.check(core.lbfgsx_upload(h, L.VEC_D, _vp(d)))
        L.check(core.lbfgsx_ls_begin(h))
        L.check(core.lbfgsx_trial_point(h, step))
        L.check(core.lbfgsx_upload(h, L.VEC_GT, _vp(gt)))
        L.check(core.lbfgsx_ls_end(h, 0))
        x = np.empty(n, dt)
        L.check(core.lbfgsx_download(h, L.VEC_X, _vp(x)))
        np.testing.assert_array_equal(x, xp + dt(step) * d)

        scal, sd, gd = np.zeros(7), np.zeros(2 * m), np.zeros(2 * m)
        L.check(core.lbfgsx_gs_post_linesearch(h, _pd(scal), _pd(sd), _pd(gd), None))
        s, y = x - xp, gt - gp  # in T, as LBFGS.h:159-160
        f8 = lambda v: v.astype(np.float64)
        want = [f8(gt) @ f8(gt), f8(x) @ f8(x), f8(s) @ f8(y), f8(y) @ f8(y), f8(s) @ f8(s), f8(gt) @ f8(s), f8(gt) @ f8(y)]
        rt = 1e-12
        for got, w, a, b in zip(scal, want, (gt, x, s, y, s, gt, gt), (gt, x, y, y, s, s, y)):
            assert abs(got - w) <= rt * (np.abs(f8(a)) @ np.abs(f8(b))) + 1e-300
        for j in range(cn):
            sj, yj = f8(S[slot_pair[j]]), f8(Y[slot_pair[j]])
            for got, u, v in ((sd[j], sj, s), (sd[m + j], yj, s), (gd[j], sj, gt), (gd[m + j], yj, gt)):
                assert abs(got - u @ f8(v)) <= rt * (np.abs(u) @ np.abs(f8(v)))

        # the pair was written into the spare column: commit it and read the history back
        L.check(core.lbfgsx_commit_correction(h))
        nc2 = min(npairs + 1, m)
        Sd, Yd = np.zeros((nc2, n), dt), np.zeros((nc2, n), dt)
        ncorr, ptr, theta = C.c_int(), C.c_int(), C.c_double()
        L.check(core.lbfgsx_bfgs_download_history(h, _vp(Sd), _vp(Yd), C.byref(ncorr), C.byref(ptr), C.byref(theta)))
        loc = npairs % m
        np.testing.assert_array_equal(Sd[loc], s)
        np.testing.assert_array_equal(Yd[loc], y)
        assert ncorr.value == nc2 and ptr.value == loc + 1
        assert theta.value == float(dt(dt(scal[3]) / dt(scal[2])))

        # d = cg g + sum coef_k b_k over the slots now stored
        slot_pair[loc] = npairs
        Sall = {**{j: S[k] for j, k in slot_pair.items() if k < npairs}, loc: s}
        Yall = {**{j: Y[k] for j, k in slot_pair.items() if k < npairs}, loc: y}
        coef = rng.standard_normal(2 * m)
        cg = -0.8
        dg = C.c_double()
        L.check(core.lbfgsx_gs_direction(h, _pd(coef), cg, C.byref(dg)))
        dd = np.empty(n, dt)
        L.check(core.lbfgsx_download(h, L.VEC_D, _vp(dd)))
        ref = cg * f8(gt)
        mag = abs(cg) * np.abs(f8(gt))
        for j in range(nc2):
            cs, cy = float(dt(coef[j])), float(dt(coef[m + j]))
            ref = ref + cs * f8(Sall[j]) + cy * f8(Yall[j])
            mag = mag + abs(cs) * np.abs(f8(Sall[j])) + abs(cy) * np.abs(f8(Yall[j]))
        eps = np.finfo(dt).eps
        assert np.all(np.abs(f8(dd) - ref) <= 4 * (nc2 + 1) * eps * mag + 1e-300)
        assert abs(dg.value - f8(gt) @ f8(dd)) <= 1e-6 * (np.abs(f8(gt)) @ np.abs(f8(dd))) if dtype == O.F32 else \
            abs(dg.value - f8(gt) @ f8(dd)) <= 1e-12 * (np.abs(f8(gt)) @ np.abs(f8(dd)))
    finally:
        core.lbfgsx_destroy(h)


@pytest.mark.parametrize("n,m,iters,ls", [(1000, 6, 3, "mt"), (1000, 6, 9, "mt"), (4098, 3, 12, "nw"), (20000, 10, 25, "mt"),
                                          (512, 1, 6, "mt"), (2000, 24, 30, "mt")])
def test_direction_equals_vector_two_loop_on_the_same_history(A, n, m, iters, ls):
    """Run `iters` iterations in Gram-space mode; the search direction of the last one (built from g_{K-1} and the K-1
    committed pairs, wrap-around included) must equal the vector two-loop applied to that same gradient and history."""
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    par = A.LBFGSParam(m=m, epsilon=0.0, epsilon_rel=0.0, max_iterations=iters)
    sv = A.LBFGSSolver(par, linesearch=A.LS_MORE_THUENTE if ls == "mt" else A.LS_NOCEDAL_WRIGHT)
    sv.set_recursion(L.RECURSION_GRAM_SPACE)
    x = O.rosen_x0(n)
    niter, fx = sv.minimize(A.ExtendedRosenbrock(), x)
    assert niter == iters
    h = sv.ctx
    d_gram = np.empty(n)
    L.check(core.lbfgsx_download(h, L.VEC_D, _vp(d_gram)))
    dg = C.c_double()
    L.check(core.lbfgsx_apply_Hv(h, L.VEC_GP, -1.0, C.byref(dg)))
    d_vec = np.empty(n)
    L.check(core.lbfgsx_download(h, L.VEC_D, _vp(d_vec)))
    assert core.lbfgsx_bfgs_ncorr(h) == min(iters - 1, m)
    assert np.linalg.norm(d_gram - d_vec) <= 1e-9 * np.linalg.norm(d_vec)


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-8), (np.float32, 2e-3)])
def test_quadratic_run_tracks_the_vector_form(A, dtype, tol):
    """Convex quadratic (cfg2 shape, small n): L-BFGS is contractive there, so the two forms stay together to rounding."""
    from lbfgspp_amd import _lib as L
    n, m = 30000, 10
    rng = np.random.default_rng(5)
    a = (1.0 + 9.0 * np.arange(n) / (n - 1)).astype(dtype)
    b = (a * (4.0 * rng.random(n) - 2.0)).astype(dtype)
    out = {}
    for form in (L.RECURSION_VECTOR, L.RECURSION_GRAM_SPACE):
        par = A.LBFGSParam(m=m, epsilon=0.0, epsilon_rel=0.0, max_iterations=25)
        sv = A.LBFGSSolver(par, linesearch=A.LS_NOCEDAL_WRIGHT, dtype=dtype)
        sv.set_recursion(form)
        tr = A.TraceBuffer(n, cap=256, stride=97)
        x = np.zeros(n, dtype)
        niter, fx = sv.minimize(A.DiagQuadratic(a, b), x, trace=tr)
        out[form] = (niter, fx, tr.fx[:tr.count].copy(), tr.xs[:tr.count].copy(), x.astype(np.float64))
    v, g = out[L.RECURSION_VECTOR], out[L.RECURSION_GRAM_SPACE]
    assert v[0] == g[0] and len(v[2]) == len(g[2])
    scale = np.abs(v[4]).max()
    assert np.abs(v[3] - g[3]).max() <= tol * scale
    assert np.abs(v[4] - g[4]).max() <= tol * scale


@pytest.mark.parametrize("dtype", [np.float64])
def test_rosenbrock_converges_like_the_vector_form(A, dtype):
    from lbfgspp_amd import _lib as L
    n, m = 10000, 6
    res = {}
    for form in (L.RECURSION_VECTOR, L.RECURSION_GRAM_SPACE):
        par = A.LBFGSParam(m=m, epsilon=1e-6, epsilon_rel=0.0, max_iterations=500)
        sv = A.LBFGSSolver(par, linesearch=A.LS_MORE_THUENTE, dtype=dtype)
        sv.set_recursion(form)
        tr = A.TraceBuffer(n, cap=600, with_x=False)
        x = O.rosen_x0(n).astype(dtype)
        niter, fx = sv.minimize(A.ExtendedRosenbrock(), x, trace=tr)
        res[form] = (niter, fx, tr.fx[:tr.count].copy(), x)
    v, g = res[L.RECURSION_VECTOR], res[L.RECURSION_GRAM_SPACE]
    # identical early trajectory (rounding-level differences only), same minimiser at the end
    k = min(12, len(v[2]), len(g[2]))
    assert np.all(np.abs(v[2][:k] - g[2][:k]) <= 1e-9 * np.abs(v[2][:k]))
    assert np.abs(g[3] - 1.0).max() < 1e-4 and g[1] < 1e-8
    assert abs(g[0] - v[0]) <= max(5, v[0] // 4)


def test_unsupported_configurations_fail_loudly(A):
    from lbfgspp_amd import _lib as L
    sv = A.LBFGSSolver(A.LBFGSParam(m=25, max_iterations=3), linesearch=A.LS_MORE_THUENTE)
    sv.set_recursion(L.RECURSION_GRAM_SPACE)
    with pytest.raises(ValueError, match="m <= 24"):
        sv.minimize(A.ExtendedRosenbrock(), O.rosen_x0(100))
    with pytest.raises(ValueError):
        sv.set_recursion(7)
    sb = A.LBFGSBSolver(A.LBFGSBParam(m=5))
    with pytest.raises(ValueError):
        sb.set_recursion(L.RECURSION_GRAM_SPACE)


# ---------------------------------------------------------------- f32 history of an f64 problem
@pytest.mark.parametrize("n,m,K", [(4099, 6, 9), (65536, 10, 12), (7, 3, 5), (1002, 24, 26)])
def test_mixed_precision_history_kernels_match_numpy(A, n, m, K):
    """Drive K line-search steps through lbfgsx_gs_post_linesearch with the history kept in f32: the stored columns are
    the float-rounded s, y; every returned dot (s, g and y rows) and the combined direction must match numpy on them."""
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    rng = np.random.default_rng(17 + n)
    h = C.c_void_p()
    L.check(core.lbfgsx_create(C.byref(h), O.F64, n, m, 0, 0))
    try:
        L.check(core.lbfgsx_gs_set_history_dtype(h, O.F32))
        slots = {}  # slot -> (s_r, y_r) as float64 views of the stored floats
        x = rng.standard_normal(n)
        g = rng.standard_normal(n)
        ptr = m
        for k in range(K):
            d = rng.standard_normal(n)
            gt = g + 1.5 * 0.3 * d + 0.01 * rng.standard_normal(n)
            L.check(core.lbfgsx_upload(h, L.VEC_X, _vp(x)))
            L.check(core.lbfgsx_upload(h, L.VEC_G, _vp(g)))
            L.check(core.lbfgsx_upload(h, L.VEC_D, _vp(d)))
            L.check(core.lbfgsx_ls_begin(h))
            L.check(core.lbfgsx_trial_point(h, 0.3))
            L.check(core.lbfgsx_upload(h, L.VEC_GT, _vp(gt)))
            L.check(core.lbfgsx_ls_end(h, 0))
            xn = x + 0.3 * d
            scal, sd, gd, yd = np.zeros(7), np.zeros(2 * m), np.zeros(2 * m), np.zeros(2 * m)
            L.check(core.lbfgsx_gs_post_linesearch(h, _pd(scal), _pd(sd), _pd(gd), _pd(yd)))
            s_r = (xn - x).astype(np.float32).astype(np.float64)
            y_r = (gt - g).astype(np.float32).astype(np.float64)
            want = [gt @ gt, xn @ xn, s_r @ y_r, y_r @ y_r, s_r @ s_r, gt @ s_r, gt @ y_r]
            for got, w in zip(scal, want):
                assert abs(got - w) <= 1e-11 * max(abs(w), 1.0) * np.sqrt(n)
            for j, (sj, yj) in slots.items():
                for got, u, v in ((sd[j], sj, s_r), (sd[m + j], yj, s_r), (gd[j], sj, gt), (gd[m + j], yj, gt),
                                  (yd[j], sj, y_r), (yd[m + j], yj, y_r)):
                    assert abs(got - u @ v) <= 1e-12 * (np.abs(u) @ np.abs(v))
            L.check(core.lbfgsx_commit_correction(h))
            loc = ptr % m
            slots[loc] = (s_r, y_r)
            ptr = loc + 1
            x, g = xn, gt
        assert core.lbfgsx_bfgs_ncorr(h) == min(K, m)
        coef = rng.standard_normal(2 * m)
        dg = C.c_double()
        L.check(core.lbfgsx_gs_direction(h, _pd(coef), -0.7, C.byref(dg)))
        dd = np.empty(n)
        L.check(core.lbfgsx_download(h, L.VEC_D, _vp(dd)))
        ref, mag = -0.7 * g, 0.7 * np.abs(g)
        for j, (sj, yj) in slots.items():
            ref = ref + coef[j] * sj + coef[m + j] * yj
            mag = mag + abs(coef[j]) * np.abs(sj) + abs(coef[m + j]) * np.abs(yj)
        assert np.all(np.abs(dd - ref) <= 4 * (2 * m + 1) * np.finfo(np.float64).eps * mag)
        assert abs(dg.value - g @ dd) <= 1e-12 * (np.abs(g) @ np.abs(dd))
        # the T-typed history is not maintained in this mode: the vector-form entry points refuse
        with pytest.raises(ArithmeticError):
            L.check(core.lbfgsx_apply_Hv(h, L.VEC_G, -1.0, C.byref(dg)))
        with pytest.raises(ArithmeticError):
            L.check(core.lbfgsx_post_linesearch(h, C.byref(dg), C.byref(dg), C.byref(dg), C.byref(dg)))
        # and the mode cannot be changed with pairs stored
        with pytest.raises(ArithmeticError):
            L.check(core.lbfgsx_gs_set_history_dtype(h, O.F64))
    finally:
        core.lbfgsx_destroy(h)


def test_mixed_precision_history_runs_converge_like_the_vector_form(A):
    from lbfgspp_amd import _lib as L
    n, m = 20000, 8
    rng = np.random.default_rng(11)
    a = 1.0 + 9.0 * np.arange(n) / (n - 1)
    b = a * (4.0 * rng.random(n) - 2.0)
    sols = {}
    for form in (L.RECURSION_VECTOR, L.RECURSION_GRAM_SPACE_F32H):
        sv = A.LBFGSSolver(A.LBFGSParam(m=m, epsilon=1e-7, epsilon_rel=0.0, max_iterations=300), linesearch=A.LS_NOCEDAL_WRIGHT)
        sv.set_recursion(form)
        x = np.zeros(n)
        niter, fx = sv.minimize(A.DiagQuadratic(a, b), x)
        sols[form] = (niter, fx, x)
    v, w = sols[L.RECURSION_VECTOR], sols[L.RECURSION_GRAM_SPACE_F32H]
    assert np.abs(w[2] - b / a).max() <= 1e-6 and np.abs(v[2] - w[2]).max() <= 1e-6
    assert abs(w[0] - v[0]) <= max(6, v[0] // 3)

    sv = A.LBFGSSolver(A.LBFGSParam(m=6, epsilon=1e-6, epsilon_rel=0.0, max_iterations=500), linesearch=A.LS_MORE_THUENTE)
    sv.set_recursion(L.RECURSION_GRAM_SPACE_F32H)
    x = O.rosen_x0(10000)
    niter, fx = sv.minimize(A.ExtendedRosenbrock(), x)
    assert fx < 1e-8 and np.abs(x - 1.0).max() < 1e-4 and niter < 400
    # the same solver object goes back to the bit-parity form afterwards
    sv.set_recursion(L.RECURSION_VECTOR)
    x1 = O.rosen_x0(10000)
    n1, f1 = sv.minimize(A.ExtendedRosenbrock(), x1)
    fresh = A.LBFGSSolver(A.LBFGSParam(m=6, epsilon=1e-6, epsilon_rel=0.0, max_iterations=500), linesearch=A.LS_MORE_THUENTE)
    x2 = O.rosen_x0(10000)
    n2, f2 = fresh.minimize(A.ExtendedRosenbrock(), x2)
    assert (n1, f1) == (n2, f2) and np.array_equal(x1, x2)
    # f32 problems have nothing to halve
    sf = A.LBFGSSolver(A.LBFGSParam(m=5), dtype=np.float32)
    with pytest.raises(ValueError):
        sf.set_recursion(L.RECURSION_GRAM_SPACE_F32H)
