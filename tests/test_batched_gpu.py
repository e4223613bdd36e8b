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
                                            # problems longer than one CU's registers: 2, 3, 8 and 16 blocks per problem with the
                                            # sums exchanged between them at every step
                                            (np.float32, 200000, 10, 0.0), (np.float64, 150016, 6, 0.0),
                                            (np.float32, 800000, 4, 0.0), (np.float64, 60000, 5, 1e-3), (np.float32, 1600000, 3, 0.0),
                                            # run far past convergence (eps < 0: 600 iterations asked for): searches that give
                                            # up, steps at rounding level, pairs the curvature test rejects
                                            (np.float32, 4096, 5, -1.0), (np.float64, 512, 3, -1.0)])
def test_one_launch_iteration_equals_statement_wise_launches(monkeypatch, dtype, n, m, eps):
    """lbfgsx_bat_iterate (post statements + recursion + first trial of the next search in ONE launch per lock-step iteration,
    the direction resident on the CU) against the three statement-wise forms it replaces: identical records and iterates.
    eps > 0: problems converge at different iterations, so finished problems sit out of later launches."""
    import lbfgspp_amd as A
    from lbfgspp_amd import batched as B
    par = A.LBFGSParam(m=m, epsilon=max(eps, 0.0), epsilon_rel=0.0, max_iterations=(400 if eps > 0 else 600 if eps < 0 else m + 6))
    got = {}
    for name, it, hv, asy in (("iterate", "1", "1", "0"), ("apply_Hv", "0", "1", "0"), ("steps", "0", "0", "0"),
                              ("async", "1", "1", "1")):
        monkeypatch.setenv("LBFGSX_BAT_FUSED_ITER", it)
        monkeypatch.setenv("LBFGSX_BAT_FUSED_HV", hv)
        monkeypatch.setenv("LBFGSX_BAT_ASYNC_TRIALS", asy)  # problems still searching ride the next launch (trial-only mode)
        got[name] = B.solve_local_lockstep(par, n, first=3, count=7, seed_base=5, dtype=dtype, return_x=True)
    for name in ("apply_Hv", "steps", "async"):
        assert np.array_equal(got["iterate"][0], got[name][0]), name
        assert np.array_equal(got["iterate"][1], got[name][1]), name
    if eps < 0:
        print("far past convergence: niter", got["iterate"][0]["niter"], "status", got["iterate"][0]["status"])
    if eps > 0:
        assert len(set(got["iterate"][0]["niter"])) > 1, got["iterate"][0]["niter"]  # the case the parameter is there for


@pytest.mark.parametrize("dtype,n,m,count", [(np.float32, 8, 1, 1), (np.float64, 4, 2, 3), (np.float32, 260, 2, 1), (np.float64, 2, 1, 2),
                                              (np.float32, 4, 31, 5), (np.float32, 100352, 1, 1)])
def test_one_launch_iteration_at_the_edges(monkeypatch, dtype, n, m, count):
    """the smallest shapes the one-launch iteration accepts -- a single problem, one 16-byte vector per problem, m = 1 (the
    recursion is three steps), m far above what n can fill, the largest n of one block with the shortest history -- against
    the statement-wise launches and against the single-problem solver"""
    import lbfgspp_amd as A
    from lbfgspp_amd import batched as B
    par = A.LBFGSParam(m=m, epsilon=0.0, epsilon_rel=0.0, max_iterations=12)
    got = {}
    for name, it, hv in (("iterate", "1", "1"), ("steps", "0", "0")):
        monkeypatch.setenv("LBFGSX_BAT_FUSED_ITER", it)
        monkeypatch.setenv("LBFGSX_BAT_FUSED_HV", hv)
        got[name] = B.solve_local_lockstep(par, n, first=2, count=count, seed_base=9, dtype=dtype, return_x=True)
    assert np.array_equal(got["iterate"][0], got["steps"][0])
    assert np.array_equal(got["iterate"][1], got["steps"][1])
    recs = got["iterate"][0]
    s = A.LBFGSSolver(par, linesearch=A.LS_MORE_THUENTE, dtype=dtype)
    for k in range(count):
        x = O.rosen_x0(n, 9 + 2 + k, O.F32 if dtype == np.float32 else O.F64).copy()
        try:
            niter, fx = s.minimize(A.ExtendedRosenbrock(), x)
            status = 0
        except (RuntimeError, ArithmeticError, ValueError):
            status, niter, fx = s.last.status, s.last.niter, s.last.fx
        assert recs["status"][k] == status, k
        if status == 0:
            assert (recs["niter"][k], recs["nfev"][k]) == (niter, s.last.nfev), k
            assert recs["fx"][k] == fx and np.array_equal(got["iterate"][1][k], x)


def test_split_problem_whose_parts_lose_each_other_fails_loudly_and_recovers(A, monkeypatch):
    """a problem split over several blocks exchanges its partial sums at every step; a part that waits in vain gives up, stops
    publishing and poisons the launch's error word, so its siblings give up too and part 0 reports the launch as failed:
    lbfgsx_bat_iterate answers LBFGSX_E_RUNTIME instead of sums.  LBFGSX_BAT_DEBUG_XCH_FAULT=k makes the k-th such launch of a
    batch look like that (part 1 of every problem mute from the start).  The batch stays usable: the next minimisation on it
    equals an undisturbed one."""
    from lbfgspp_amd import batched as B
    par = A.LBFGSParam(m=4, epsilon=0.0, epsilon_rel=0.0, max_iterations=8)
    n, count = 200000, 3  # f32: two blocks per problem
    want, xw = B.solve_local_lockstep(par, n, first=1, count=count, seed_base=21, dtype=np.float32, return_x=True)
    monkeypatch.setenv("LBFGSX_BAT_DEBUG_XCH_FAULT", "3")
    batch = B.LockstepBatch(par, n, count, dtype=np.float32, device=0)
    monkeypatch.delenv("LBFGSX_BAT_DEBUG_XCH_FAULT")
    with pytest.raises(RuntimeError, match="exchange timed out"):
        batch.minimize(first=1, seed_base=21, return_x=True)
    got, xg = batch.minimize(first=1, seed_base=21, return_x=True)
    assert batch.stats["fused"]
    batch.close()
    assert np.array_equal(got, want) and np.array_equal(xg, xw)


def test_lockstep_batch_history_limit_is_reported(A):
    """the lock-step batch's descriptors carry 32 column ids (include/lbfgsx.h, LBFGSX_MAX_M_BATCH = 31): m = 31 runs (the 98-slot
    case above), m = 32 is refused with a message that names the limit and the form that has none"""
    from lbfgspp_amd import batched as B
    par = A.LBFGSParam(m=32, epsilon=0.0, epsilon_rel=0.0, max_iterations=3)
    with pytest.raises(ValueError, match="m <= 31"):
        B.solve_local_lockstep(par, 4096, first=0, count=2, seed_base=1, dtype=np.float32)
    recs = B.solve_local(par, A.ExtendedRosenbrock.objective, 4096, 0, 2, dtype=np.float32, nthreads=2)  # concurrent contexts: any m
    assert list(recs["niter"]) == [3, 3]


def test_resident_batch_is_reused_across_minimisations(A):
    """lbfgsx_lockstep_create / _minimize: the batch allocated once; a second minimisation of the same ids repeats the first
    bit for bit, other ids give other problems, and both equal the one-shot call; the stats say which form ran"""
    from lbfgspp_amd import batched as B
    par = A.LBFGSParam(m=5, epsilon=0.0, epsilon_rel=0.0, max_iterations=9)
    n, count = 20000, 6
    batch = B.LockstepBatch(par, n, count, dtype=np.float32, device=0)
    a, xa = batch.minimize(first=0, seed_base=11, return_x=True)
    assert batch.stats["fused"] and batch.stats["lockstep_iterations"] >= 10 and batch.stats["kernel_ms"] == 0.0
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


def test_one_launch_iteration_takes_the_curvature_decision_of_the_reference(A):
    """lbfgsx_bat_iterate on hand-made states (no objective involved): the post statements of three problems whose new pair
    is (0) usable, (1) flat -- y = 0, s.y = 0 <= eps y.y -- and (2) of negative curvature.  The kernel must store the pair
    and report its sums in every case, but build the direction from "history + pair" only for problem 0 (LBFGS.h:161,
    BFGSMat.h:83-97): the others get -H g of the history they had (none here: -g), exactly what the host concludes from the
    same sums.  A second launch on top of a stored pair checks the two-loop coefficients against numpy."""
    core, _ = A.load()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    H2D, D2H = 1, 2

    class ItDesc(C.Structure):
        _fields_ = [("active", C.c_int), ("flags", C.c_int), ("cur", C.c_int), ("xp", C.c_int), ("trial", C.c_int),
                    ("ncorr", C.c_int), ("spare", C.c_int), ("pad", C.c_int), ("step", C.c_double), ("pcol", C.c_int * 32)]
    n, m, P = 8192, 4, 3
    bat = C.c_void_p()
    core.lbfgsx_bat_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int]
    core.lbfgsx_bat_vec.restype = C.c_void_p
    core.lbfgsx_bat_vec.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    core.lbfgsx_bat_iterate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    core.lbfgsx_bat_iterate_ok.argtypes = [C.c_void_p]
    core.lbfgsx_bat_destroy.argtypes = [C.c_void_p]
    core.lbfgsx_bat_sync.argtypes = [C.c_void_p]
    L = __import__("lbfgspp_amd")._lib
    L.check(core.lbfgsx_bat_create(C.byref(bat), L.F64, n, m, P, 0))
    try:
        assert core.lbfgsx_bat_iterate_ok(bat) == 1
        rng = np.random.default_rng(3)

        def put(kind, point, p, v):
            v = np.ascontiguousarray(v, dtype=np.float64)
            assert hip.hipMemcpy(core.lbfgsx_bat_vec(bat, kind, point, p), v.ctypes.data_as(C.c_void_p), v.nbytes, H2D) == 0

        def get(kind, point, p):
            v = np.empty(n)
            L.check(core.lbfgsx_bat_sync(bat))
            assert hip.hipMemcpy(v.ctypes.data_as(C.c_void_p), core.lbfgsx_bat_vec(bat, kind, point, p), v.nbytes, D2H) == 0
            return v
        xp = [rng.standard_normal(n) for _ in range(P)]
        gp = [rng.standard_normal(n) for _ in range(P)]
        s = [0.1 * rng.standard_normal(n) for _ in range(P)]
        y = [2.0 * s[0] + 0.01 * rng.standard_normal(n), np.zeros(n), -1.5 * s[2]]
        x = [xp[p] + s[p] for p in range(P)]
        g = [gp[p] + y[p] for p in range(P)]
        for p in range(P):
            put(0, 0, p, xp[p]); put(1, 0, p, gp[p]); put(0, 1, p, x[p]); put(1, 1, p, g[p])
        desc = (ItDesc * P)()
        for p in range(P):
            desc[p].active, desc[p].flags, desc[p].cur, desc[p].xp, desc[p].ncorr, desc[p].spare = 1, 1, 1, 0, 0, m  # POST
        out = np.zeros(P * 8)
        L.check(core.lbfgsx_bat_iterate(bat, L.OBJ_EXT_ROSENBROCK, desc, out.ctypes.data_as(C.c_void_p)))
        out = out.reshape(P, 8)
        eps = np.finfo(np.float64).eps
        for p in range(P):
            sp, yp = x[p] - xp[p], g[p] - gp[p]   # the kernel's s and y: differences of the stored points
            gg, xx, sy, yy = g[p] @ g[p], x[p] @ x[p], sp @ yp, yp @ yp
            assert np.allclose(out[p, :4], [gg, xx, sy, yy], rtol=1e-13, atol=1e-13)
            accept = out[p, 2] > eps * out[p, 3]
            assert accept == (p == 0)
            d = get(2, 0, p)
            if accept:   # one pair: alpha = s.q / s.y, q -= alpha y, q /= theta, q += (alpha - y.q / s.y) s
                q = -g[p]
                alpha = (sp @ q) / sy
                q = q - alpha * yp
                q = q / (yy / sy)
                q = q + (alpha - (yp @ q) / sy) * sp
                assert np.abs(d - q).max() <= 1e-12 * np.abs(q).max()
            else:
                assert np.array_equal(d, -g[p])
            assert abs(out[p, 4] - g[p] @ d) <= 1e-11 * abs(g[p] @ d)   # grad . drt
            # the pair is stored either way (the host decides whether the ring takes it)
            assert out[p, 7] == 0.0
        # second launch: problem 0 has taken its pair (column m), the others still have none; new points on top
        for p in range(P):
            put(0, 2, p, x[p] + 0.05 * s[p]); put(1, 2, p, g[p] + 0.1 * s[p])
            desc[p].cur, desc[p].xp, desc[p].spare = 2, 1, 0
            desc[p].ncorr = 1 if p == 0 else 0
            desc[p].pcol[0] = m
        out = np.zeros(P * 8)
        L.check(core.lbfgsx_bat_iterate(bat, L.OBJ_EXT_ROSENBROCK, desc, out.ctypes.data_as(C.c_void_p)))
        out = out.reshape(P, 8)
        assert all(out[p, 2] > eps * out[p, 3] for p in range(P))   # s = 0.05 s, y = 0.1 s: usable everywhere now
        d0 = get(2, 0, 0)
        # problem 0: two pairs, newest first (numpy two-loop, BFGSMat.h:276-302)
        S = [0.05 * s[0], x[0] - xp[0]]
        Y = [(g[0] + 0.1 * s[0]) - g[0], g[0] - gp[0]]
        q = -(g[0] + 0.1 * s[0])
        al = []
        for si, yi in zip(S, Y):
            a_ = (si @ q) / (si @ yi)
            al.append(a_)
            q = q - a_ * yi
        q = q / ((Y[0] @ Y[0]) / (S[0] @ Y[0]))
        for si, yi, a_ in reversed(list(zip(S, Y, al))):
            q = q + (a_ - (yi @ q) / (si @ yi)) * si
        assert np.abs(d0 - q).max() <= 1e-11 * np.abs(q).max()
    finally:
        core.lbfgsx_bat_destroy(bat)
