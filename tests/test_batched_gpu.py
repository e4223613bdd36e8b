"""-m gpu: batched mode (cfg5 shape, scaled down) -- every problem of a batch equals its stand-alone solve."""
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


def test_batched_equals_single_solves_and_oracle(oracle):
    import lbfgspp_amd as A
    from lbfgspp_amd import batched as B
    n, m, iters, count = 2048, 10, 12, 12
    par = A.LBFGSParam(m=m, epsilon=0.0, epsilon_rel=0.0, max_iterations=iters)
    recs = B.solve_local(par, A.ExtendedRosenbrock.objective, n, first=5, count=count, seed_base=1000,
                         dtype=np.float32, nthreads=4)
    assert len(recs) == count
    s = A.LBFGSSolver(par, linesearch=A.LS_MORE_THUENTE, dtype=np.float32)
    p = O.lbfgs_params(m=m, epsilon=0, epsilon_rel=0, max_iterations=iters)
    for k in range(count):
        x0 = O.rosen_x0(n, 1000 + 5 + k, O.F32)
        x = x0.copy()
        try:
            niter, fx = s.minimize(A.ExtendedRosenbrock(), x)
            status = 0
        except (RuntimeError, ArithmeticError, ValueError):
            status, niter, fx = s.last.status, s.last.niter, s.last.fx
        assert (recs["niter"][k], recs["nfev"][k], recs["status"][k]) == (niter, s.last.nfev, status)
        if status == 0:
            assert recs["fx"][k] == fx
            _, r = oracle.lbfgs(O.F32, O.LS_MT, O.OBJ_ROSEN, x0, p)
            assert r.niter == niter and abs(r.fx - fx) <= 1e-4 * max(1.0, abs(fx))


def test_batched_is_deterministic_across_thread_counts():
    import lbfgspp_amd as A
    from lbfgspp_amd import batched as B
    par = A.LBFGSParam(m=6, epsilon=0.0, epsilon_rel=0.0, max_iterations=8)
    a = B.solve_local(par, A.ExtendedRosenbrock.objective, 4096, 0, 16, dtype=np.float32, nthreads=1)
    b = B.solve_local(par, A.ExtendedRosenbrock.objective, 4096, 0, 16, dtype=np.float32, nthreads=8)
    assert np.array_equal(a, b)


# n = 4098 (not a multiple of the vector width) takes the step-wise two-loop launches, the others the one-launch
# register-resident recursion (lbfgsx_bat_apply_Hv) in its 14 / 56 / 98-slot variants (98: with the LDS overflow)
@pytest.mark.parametrize("dtype,n,m,iters,count", [(np.float32, 4098, 10, 14, 9), (np.float64, 2048, 5, 20, 5),
                                                   (np.float32, 1000, 3, 40, 17), (np.float32, 100000, 10, 7, 3),
                                                   (np.float64, 50000, 7, 6, 2), (np.float32, 30000, 4, 8, 3)])
def test_lockstep_batch_is_bit_identical_to_single_solves(dtype, n, m, iters, count):
    """every problem of the lock-step batch == LBFGSSolver<T, LineSearchMoreThuente> on the same start point,
    including problems that hit max_linesearch / converge at different iterations"""
    import lbfgspp_amd as A
    from lbfgspp_amd import batched as B
    par = A.LBFGSParam(m=m, epsilon=1e-3, epsilon_rel=0.0, max_iterations=iters, max_linesearch=6)
    recs, xs = B.solve_local_lockstep(par, n, first=3, count=count, seed_base=1000, dtype=dtype, return_x=True)
    s = A.LBFGSSolver(par, linesearch=A.LS_MORE_THUENTE, dtype=dtype)
    dt = O.F64 if dtype == np.float64 else O.F32
    for k in range(count):
        x = O.rosen_x0(n, 1000 + 3 + k, dt)
        try:
            niter, fx = s.minimize(A.ExtendedRosenbrock(), x)
            status = 0
        except (RuntimeError, ArithmeticError, ValueError):
            status, niter, fx = s.last.status, recs["niter"][k], s.last.fx
        assert recs["status"][k] == status
        if status == 0:
            assert (recs["niter"][k], recs["nfev"][k]) == (niter, s.last.nfev)
            assert recs["fx"][k] == fx and recs["gnorm"][k] == s.last.gnorm
            assert np.array_equal(xs[k], x)


def test_lockstep_equals_threaded_batch():
    import lbfgspp_amd as A
    from lbfgspp_amd import batched as B
    par = A.LBFGSParam(m=6, epsilon=0.0, epsilon_rel=0.0, max_iterations=10)
    a = B.solve_local(par, A.ExtendedRosenbrock.objective, 4096, 0, 24, dtype=np.float32, nthreads=4)
    b = B.solve_local_lockstep(par, 4096, 0, 24, dtype=np.float32)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("dtype,n", [(np.float32, 100000), (np.float64, 20000), (np.float32, 4096)])
def test_one_launch_two_loop_equals_step_wise_launches(monkeypatch, dtype, n):
    """lbfgsx_bat_apply_Hv (q resident in registers / LDS for the whole recursion) against the LBFGSX_BAT_TWOLOOP
    step launches: identical records and iterates"""
    import lbfgspp_amd as A
    from lbfgspp_amd import batched as B
    par = A.LBFGSParam(m=8, epsilon=0.0, epsilon_rel=0.0, max_iterations=11)
    monkeypatch.setenv("LBFGSX_BAT_FUSED_HV", "1")
    a, xa = B.solve_local_lockstep(par, n, first=0, count=5, seed_base=77, dtype=dtype, return_x=True)
    monkeypatch.setenv("LBFGSX_BAT_FUSED_HV", "0")
    b, xb = B.solve_local_lockstep(par, n, first=0, count=5, seed_base=77, dtype=dtype, return_x=True)
    assert np.array_equal(a, b)
    for u, v in zip(xa, xb):
        assert np.array_equal(u, v)


@pytest.mark.parametrize("dtype,n,m,eps", [(np.float32, 100000, 10, 0.0), (np.float64, 50176, 6, 0.0), (np.float32, 4096, 3, 0.0),
                                            (np.float64, 2048, 5, 1e-3), (np.float32, 100352, 31, 0.0),
                                            # problems longer than one CU's registers: 2, 3 and 8 blocks per problem with the
                                            # sums exchanged between them at every step
                                            (np.float32, 200000, 10, 0.0), (np.float64, 150016, 6, 0.0),
                                            (np.float32, 800000, 4, 0.0), (np.float64, 60000, 5, 1e-3)])
def test_one_launch_iteration_equals_statement_wise_launches(monkeypatch, dtype, n, m, eps):
    """lbfgsx_bat_iterate (post statements + recursion + first trial of the next search in ONE launch per lock-step iteration,
    the direction resident on the CU) against the three statement-wise forms it replaces: identical records and iterates.
    eps > 0: problems converge at different iterations, so finished problems sit out of later launches."""
    import lbfgspp_amd as A
    from lbfgspp_amd import batched as B
    par = A.LBFGSParam(m=m, epsilon=eps, epsilon_rel=0.0, max_iterations=(400 if eps > 0 else m + 6))
    got = {}
    for name, it, hv in (("iterate", "1", "1"), ("apply_Hv", "0", "1"), ("steps", "0", "0")):
        monkeypatch.setenv("LBFGSX_BAT_FUSED_ITER", it)
        monkeypatch.setenv("LBFGSX_BAT_FUSED_HV", hv)
        got[name] = B.solve_local_lockstep(par, n, first=3, count=7, seed_base=5, dtype=dtype, return_x=True)
    for name in ("apply_Hv", "steps"):
        assert np.array_equal(got["iterate"][0], got[name][0]), name
        assert np.array_equal(got["iterate"][1], got[name][1]), name
    if eps > 0:
        assert len(set(got["iterate"][0]["niter"])) > 1, got["iterate"][0]["niter"]  # the case the parameter is there for


def test_resident_batch_is_reused_across_minimisations(A):
    """lbfgsx_lockstep_create / _minimize: the batch allocated once; a second minimisation of the same ids repeats the first
    bit for bit, other ids give other problems, and both equal the one-shot call; the stats say which form ran"""
    from lbfgspp_amd import batched as B
    par = A.LBFGSParam(m=5, epsilon=0.0, epsilon_rel=0.0, max_iterations=9)
    n, count = 20000, 6
    batch = B.LockstepBatch(par, n, count, dtype=np.float32, device=0)
    a, xa = batch.minimize(first=0, seed_base=11, return_x=True)
    assert batch.stats["fused"] and batch.stats["lockstep_iterations"] == 10 and batch.stats["kernel_ms"] == 0.0
    b, xb = batch.minimize(first=10, seed_base=11, return_x=True)
    batch.set_timing(True)
    c, xc = batch.minimize(first=0, seed_base=11, return_x=True)
    st = batch.stats
    batch.close()
    assert np.array_equal(a, c) and np.array_equal(xa, xc)
    assert not np.array_equal(xa, xb)
    one, xo = B.solve_local_lockstep(par, n, first=10, count=count, seed_base=11, dtype=np.float32, return_x=True)
    assert np.array_equal(b, one) and np.array_equal(xb, xo)
    # 1 evaluation + one launch per lock-step iteration + the further trials; every launch with results is waited for once
    assert st["kernel_ms"] > 0.0 and st["launches"] >= 11 and st["waits"] == st["launches"] and st["wait_timeouts"] == 0


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], None])
def test_multi_device_batch_equals_the_single_device_batch(devices):
    """SURVEY 8(e) behind the C ABI: lbfgsx_batch_minimize_lockstep_multi gives every listed device a contiguous block of
    problem ids on its own host thread.  On the one-GPU test box the list repeats device 0 (None: every device of the box);
    records and iterates are bit-identical to the single-device call, uneven blocks included."""
    import lbfgspp_amd as A
    from lbfgspp_amd import batched as B
    if devices is None:
        devices = list(range(A.load()[0].lbfgsx_device_count()))
    n, m, iters, count = 30000, 6, 9, 11
    par = A.LBFGSParam(m=m, epsilon=1e-3, epsilon_rel=0.0, max_iterations=iters, max_linesearch=6)
    r1, x1 = B.solve_local_lockstep(par, n, first=2, count=count, seed_base=1000, dtype=np.float32, return_x=True)
    r2, x2 = B.solve_local_lockstep(par, n, first=2, count=count, seed_base=1000, dtype=np.float32, return_x=True,
                                    devices=devices)
    assert np.array_equal(r1, r2) and np.array_equal(x1, x2)
    # blocks follow shard_range: the records of block r start at its first id
    f, c = B.shard_range(count, len(devices) - 1, len(devices))
    r3 = B.solve_local_lockstep(par, n, first=2 + f, count=c, seed_base=1000, dtype=np.float32)
    assert np.array_equal(r3, r1[f:f + c])


def test_multi_device_batch_reports_bad_device_lists():
    import lbfgspp_amd as A
    from lbfgspp_amd import batched as B
    par = A.LBFGSParam(m=4, epsilon=0.0, epsilon_rel=0.0, max_iterations=3)
    with pytest.raises(ValueError):
        B.solve_local_lockstep(par, 1024, 0, 4, devices=[])
    with pytest.raises((ValueError, RuntimeError)):
        B.solve_local_lockstep(par, 1024, 0, 4, devices=[0, 99])


def test_rccl_allgather_of_the_result_records():
    """lbfgsx_rccl_allgather_records: the one exchange step of the sharded batch natively over RCCL (ncclCommInitAll +
    a grouped ncclAllGather, loaded with dlopen).  On the one-GPU box the communicator has a single rank: the records of
    a real batch come back from the device unchanged and in problem-id order; on a box with more GPUs every device must
    hold the same full array.  A device listed twice is refused (one rank per GPU)."""
    import ctypes as C
    import lbfgspp_amd as A
    from lbfgspp_amd import batched as B
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    par = A.LBFGSParam(m=4, epsilon=0.0, epsilon_rel=0.0, max_iterations=5)
    recs = B.solve_local_lockstep(par, 2048, first=0, count=37, dtype=np.float32)
    raw = np.ascontiguousarray(recs).view(np.uint8).reshape(37, -1)
    ndev = core.lbfgsx_device_count()
    devs = (C.c_int * ndev)(*range(ndev))
    outp = (C.c_void_p * ndev)()
    L.check(core.lbfgsx_rccl_allgather_records(devs, ndev, raw.ctypes.data_as(C.c_void_p), 37, raw.shape[1], outp))
    for r in range(ndev):
        back = np.zeros_like(raw)
        L.check(core.lbfgsx_device_download(r, outp[r], raw.size, back.ctypes.data_as(C.c_void_p)))
        core.lbfgsx_device_free(r, outp[r])
        assert np.array_equal(back, raw)
    dup = (C.c_int * 2)(0, 0)
    out2 = (C.c_void_p * 2)()
    with pytest.raises(ValueError, match="listed twice"):
        L.check(core.lbfgsx_rccl_allgather_records(dup, 2, raw.ctypes.data_as(C.c_void_p), 37, raw.shape[1], out2))


@pytest.mark.parametrize("ls,obj,dtype,n,m,iters", [("nw", "rosen", np.float64, 20000, 6, 25), ("mt", "quad", np.float64, 30001, 5, 20),
                                                    ("nw", "quad", np.float32, 10000, 4, 15), ("nw", "rosen", np.float32, 4096, 5, 20)])
def test_lockstep_batch_with_another_line_search_or_objective_equals_single_solves(A, ls, obj, dtype, n, m, iters):
    """the lock-step batch was L-BFGS + More-Thuente + extended Rosenbrock only.  The line search is now the
    template parameter the reference's solver has (LBFGS.h:20-21) for the two policies that exist as state machines, and
    the built-in objective is selectable.  Every batch member must be the single-problem solver's run on the same data,
    bit for bit -- also when a member's search fails (Nocedal-Wright in f32 runs out of precision: same status, same
    message class, same evaluation count as the stand-alone solve that throws)."""
    from lbfgspp_amd import batched as B
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    LS = L.LS_NOCEDAL_WRIGHT if ls == "nw" else L.LS_MORE_THUENTE
    OBJ = L.OBJ_DIAG_QUAD if obj == "quad" else L.OBJ_EXT_ROSENBROCK
    par = A.LBFGSParam(m=m, epsilon=0.0, epsilon_rel=0.0, max_iterations=iters)
    P, seed = 7, 300
    recs, xs = B.solve_local_lockstep(par, n, 0, P, seed_base=seed, dtype=dtype, return_x=True, linesearch=LS, objective=OBJ, kappa=10.0)
    failed = 0
    for p in range(P):
        s = A.LBFGSSolver(par, linesearch=LS, dtype=dtype)
        ctx = s.prepare(n)
        if obj == "quad":
            L.check(core.lbfgsx_gen_diag_quad(ctx, 10.0, seed + p))
            L.check(core.lbfgsx_fill(ctx, L.VEC_X, 0.0))
            f = A.DiagQuadratic()
        else:
            L.check(core.lbfgsx_gen_rosen_x0(ctx, seed + p))
            f = A.ExtendedRosenbrock()
        status = 0
        try:
            niter, fx = s.minimize_resident(f, n)
        except RuntimeError:
            status = L.E_RUNTIME
            failed += 1
        x = np.empty(n, dtype)
        assert recs["status"][p] == status
        assert recs["nfev"][p] == s.last.nfev
        if status == 0:
            L.check(core.lbfgsx_download(ctx, L.VEC_X, x.ctypes.data_as(C.c_void_p)))
            assert (recs["niter"][p], recs["fx"][p]) == (niter, fx) and recs["gnorm"][p] == s.last.gnorm
            assert np.array_equal(xs[p], x)
        s.close()
    assert failed < P   # the comparison above is not vacuous


def test_lockstep_batch_refuses_policies_without_a_state_machine(A):
    from lbfgspp_amd import batched as B
    from lbfgspp_amd import _lib as L
    par = A.LBFGSParam(m=4, epsilon=0.0, epsilon_rel=0.0, max_iterations=3)
    with pytest.raises(ValueError, match="state machines"):
        B.solve_local_lockstep(par, 1024, 0, 4, linesearch=L.LS_BACKTRACKING)
    with pytest.raises(ValueError, match="objective"):
        B.solve_local_lockstep(par, 1024, 0, 4, objective=7)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_threaded_batch_of_lbfgsb_problems_equals_single_solves_and_is_deterministic(A, dtype):
    """Batched L-BFGS-B (lbfgsx_batch_minimize with LBFGSX_ALGO_LBFGSB): independent box-constrained problems, each on its own
    context and stream, `nthreads` of them resident at a time -- while one problem's host algebra or Cauchy chain runs, the
    others' kernels fill the GPU.  Not a lock-step kernel batch (the reference solves one problem per call, LBFGSB.h:116-262;
    DESIGN 4c): every problem is the stand-alone solve, so the records are identical whatever the thread count, and equal
    to LBFGSBSolver::minimize on the same generated instance -- with every per-context mechanism of the bounded path
    (kept compact copy, carried Gram, the passes launched ahead) running concurrently in several contexts."""
    from lbfgspp_amd import batched as B
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    n, m, iters, count, first, seed = 60000, 8, 25, 10, 3, 700
    par = A.LBFGSBParam(m=m, epsilon=0.0, epsilon_rel=0.0, past=0, max_iterations=iters)
    r1 = B.solve_local(par, A.DiagQuadratic.objective, n, first, count, seed_base=seed, algo=L.ALGO_LBFGSB, dtype=dtype, nthreads=1)
    r6 = B.solve_local(par, A.DiagQuadratic.objective, n, first, count, seed_base=seed, algo=L.ALGO_LBFGSB, dtype=dtype, nthreads=6)
    assert np.array_equal(r1, r6)
    assert (r1["status"] == 0).all() and (r1["niter"] == iters).all()
    s = A.LBFGSBSolver(par, dtype=dtype)
    for k in (0, 4, 9):
        ctx = s.prepare(n)
        L.check(core.lbfgsx_gen_diag_quad(ctx, 10.0, seed + first + k))
        L.check(core.lbfgsx_fill(ctx, L.VEC_X, 0.0))
        L.check(core.lbfgsx_fill(ctx, L.VEC_LB, -1.0))
        L.check(core.lbfgsx_fill(ctx, L.VEC_UB, 1.0))
        niter, fx = s.minimize_resident(A.DiagQuadratic(), n)
        assert (r1["niter"][k], r1["nfev"][k]) == (niter, s.last.nfev) and r1["fx"][k] == fx
    s.close()
