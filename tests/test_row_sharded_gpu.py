"""-m gpu: one problem row-sharded over several ranks (LBFGSSolver::set_reducer).  The ranks are
emulated by threads of this process -- one solver / context / stream each on the same GPU -- and the all-reduce by a
barrier: what is checked is the sharding logic itself (every n-length sum passes through the reducer before any scalar
decision; the generators produce the right slice), independent of the transport bench.py uses (RCCL)."""
import ctypes as C
import threading

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


class ThreadAllReduce:
    """sum over `world` threads, identical result (same order of additions) in every thread"""

    def __init__(self, world):
        self.world, self.slots, self.calls = world, [None] * world, [0] * world
        self.barrier = threading.Barrier(world, timeout=60)

    def reducer(self, rank):
        def red(v):
            self.calls[rank] += 1
            self.slots[rank] = v.copy()
            self.barrier.wait()
            total = self.slots[0].copy()
            for r in range(1, self.world):
                total += self.slots[r]
            self.barrier.wait()
            v[:] = total
        return red


def _run_sharded(A, obj, n, bounds, m, iters, ls, form, dtype=np.float64):
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    world = len(bounds) - 1
    ar = ThreadAllReduce(world)
    out = [None] * world
    err = []

    def worker(rank):
        try:
            lo, hi = bounds[rank], bounds[rank + 1]
            sv = A.LBFGSSolver(A.LBFGSParam(m=m, epsilon=0.0, epsilon_rel=0.0, max_iterations=iters), linesearch=ls, dtype=dtype)
            sv.set_recursion(form)
            sv.set_reducer(ar.reducer(rank))
            ctx = sv.prepare(hi - lo)
            L.check(core.lbfgsx_set_shard(ctx, lo, n))
            if obj == "rosen":
                L.check(core.lbfgsx_gen_rosen_x0(ctx, 7))
                f = A.ExtendedRosenbrock()
            else:
                L.check(core.lbfgsx_gen_diag_quad(ctx, 10.0, 1))
                L.check(core.lbfgsx_fill(ctx, L.VEC_X, 0.0))
                f = A.DiagQuadratic()
            tr = A.TraceBuffer(hi - lo, cap=256, with_x=False)
            niter, fx = sv.minimize_resident(f, hi - lo, trace=tr)
            x = np.empty(hi - lo, dtype)
            L.check(core.lbfgsx_download(sv.ctx, L.VEC_X, x.ctypes.data_as(C.c_void_p)))
            out[rank] = (niter, fx, tr.fx[:tr.count].copy(), x, sv.last.nfev)
            sv.close()
        except BaseException as e:  # noqa: B902  (a failed rank must not leave the others at the barrier)
            err.append(e)
            ar.barrier.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not err, err
    return out, ar


def _run_single(A, obj, n, m, iters, ls, form, dtype=np.float64):
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    sv = A.LBFGSSolver(A.LBFGSParam(m=m, epsilon=0.0, epsilon_rel=0.0, max_iterations=iters), linesearch=ls, dtype=dtype)
    sv.set_recursion(form)
    ctx = sv.prepare(n)
    if obj == "rosen":
        L.check(core.lbfgsx_gen_rosen_x0(ctx, 7))
        f = A.ExtendedRosenbrock()
    else:
        L.check(core.lbfgsx_gen_diag_quad(ctx, 10.0, 1))
        L.check(core.lbfgsx_fill(ctx, L.VEC_X, 0.0))
        f = A.DiagQuadratic()
    tr = A.TraceBuffer(n, cap=256, with_x=False)
    niter, fx = sv.minimize_resident(f, n, trace=tr)
    x = np.empty(n, dtype)
    L.check(core.lbfgsx_download(sv.ctx, L.VEC_X, x.ctypes.data_as(C.c_void_p)))
    r = (niter, fx, tr.fx[:tr.count].copy(), x, sv.last.nfev)
    sv.close()
    return r


@pytest.mark.parametrize("obj,n,bounds,m,iters", [("rosen", 40000, (0, 20000, 40000), 6, 14),
                                                  ("rosen", 30002, (0, 1000, 17002, 30002), 4, 11),
                                                  ("quad", 50001, (0, 12345, 50001), 10, 25)])
def test_row_sharded_run_equals_the_single_device_run(A, obj, n, bounds, m, iters):
    from lbfgspp_amd import _lib as L
    ls = A.LS_MORE_THUENTE if obj == "rosen" else A.LS_NOCEDAL_WRIGHT
    single = _run_single(A, obj, n, m, iters, ls, L.RECURSION_GRAM_SPACE)
    shards, ar = _run_sharded(A, obj, n, bounds, m, iters, ls, L.RECURSION_GRAM_SPACE)
    world = len(bounds) - 1
    # every rank took the same decisions and saw the same (global) objective values
    for r in range(1, world):
        assert shards[r][0] == shards[0][0] and shards[r][4] == shards[0][4]
        assert np.array_equal(shards[r][2], shards[0][2])
    assert len(set(ar.calls)) == 1  # the same number of all-reduces everywhere
    # and they are the single-device run up to the rounding of a different split of the sums
    assert shards[0][0] == single[0] and len(shards[0][2]) == len(single[2])
    k = min(10, len(single[2]))
    assert np.all(np.abs(shards[0][2][:k] - single[2][:k]) <= 1e-9 * np.abs(single[2][:k]))
    x = np.concatenate([s[3] for s in shards])
    tol = 1e-6 if obj == "rosen" else 1e-9
    assert np.abs(x - single[3]).max() <= tol * max(1.0, np.abs(single[3]).max())


def test_row_sharded_data_generators_produce_the_slices(A):
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    n, lo, hi = 10001, 3000, 10001
    full, part = C.c_void_p(), C.c_void_p()
    L.check(core.lbfgsx_create(C.byref(full), O.F64, n, 2, 0, 0))
    L.check(core.lbfgsx_create(C.byref(part), O.F64, hi - lo, 2, 0, 0))
    try:
        L.check(core.lbfgsx_set_shard(part, lo, n))
        assert core.lbfgsx_set_shard(part, lo + 1, n) == L.E_INVALID  # would run past the end
        for h in (full, part):
            L.check(core.lbfgsx_gen_diag_quad(h, 10.0, 3))
            L.check(core.lbfgsx_gen_rosen_x0(h, 11))
        for which in (L.VEC_A, L.VEC_B, L.VEC_X):
            a, b = np.empty(n), np.empty(hi - lo)
            L.check(core.lbfgsx_download(full, which, a.ctypes.data_as(C.c_void_p)))
            L.check(core.lbfgsx_download(part, which, b.ctypes.data_as(C.c_void_p)))
            assert np.array_equal(a[lo:hi], b)
    finally:
        core.lbfgsx_destroy(full)
        core.lbfgsx_destroy(part)


def test_row_sharding_needs_the_gram_space_recursion(A):
    sv = A.LBFGSSolver(A.LBFGSParam(m=5, max_iterations=3), linesearch=A.LS_MORE_THUENTE)
    sv.set_reducer(lambda v: None)
    with pytest.raises(ValueError, match="Gram-space"):
        sv.minimize(A.ExtendedRosenbrock(), O.rosen_x0(100))
    sv.set_reducer(None)
    sv.minimize(A.ExtendedRosenbrock(), O.rosen_x0(100))
    sb = A.LBFGSBSolver(A.LBFGSBParam(m=5))
    with pytest.raises(ValueError):
        sb.set_reducer(lambda v: None)


# ---- the product's own all-reduce (lbfgsx_comm_*, csrc/rccl_allreduce.hip) and LBFGSSolver::set_devices ----------------
def _comm_local(A, devices):
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    h = C.c_void_p()
    arr = (C.c_int * len(devices))(*devices)
    L.check(core.lbfgsx_comm_create_local(C.byref(h), arr, len(devices)))
    info = (C.c_int * 4)()
    L.check(core.lbfgsx_comm_info(h, C.byref(info)))
    return h, tuple(info)


def test_native_allreduce_single_rank_goes_through_rccl(A):
    """One rank per distinct device: the communicator is RCCL's (ncclCommInitAll), and lbfgsx_comm_allreduce_sum is a real
    ncclAllReduce even with a single rank -- the bundle comes back unchanged, bit for bit."""
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    ndev = core.lbfgsx_device_count()
    comm, info = _comm_local(A, list(range(ndev)))
    assert info[0] == ndev and info[1] == ndev and info[2] == 1 and info[3] > 20000   # RCCL version code, e.g. 22xxx
    rng = np.random.default_rng(3)
    bufs = [rng.standard_normal(67) for _ in range(ndev)]
    want = np.sum(bufs, axis=0) if ndev > 1 else bufs[0].copy()
    outs = [b.copy() for b in bufs]
    th = [threading.Thread(target=lambda r=r: L.check(core.lbfgsx_comm_allreduce_sum(
        comm, r, outs[r].ctypes.data_as(C.POINTER(C.c_double)), 67))) for r in range(ndev)]
    [t.start() for t in th]
    [t.join() for t in th]
    for r in range(ndev):
        assert np.array_equal(outs[r], outs[0])
        assert np.allclose(outs[r], want, rtol=1e-15, atol=0) if ndev > 1 else np.array_equal(outs[r], want)
    assert core.lbfgsx_comm_calls(comm, 0) == 1
    big = np.zeros(513)
    assert core.lbfgsx_comm_allreduce_sum(comm, 0, big.ctypes.data_as(C.POINTER(C.c_double)), 513) == L.E_INVALID
    core.lbfgsx_comm_destroy(comm)


def test_native_allreduce_thread_emulated_ranks(A):
    """A device listed three times cannot be three RCCL ranks: the communicator adds the bundles in rank order in host
    memory behind a barrier; every rank gets the same bits, call after call; an abort releases the waiting ranks."""
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    comm, info = _comm_local(A, [0, 0, 0])
    assert info[:3] == (3, 3, 0)
    rng = np.random.default_rng(4)
    rounds = [[rng.standard_normal(31) for _ in range(3)] for _ in range(50)]
    got = [[None] * 50 for _ in range(3)]

    def worker(r):
        for k in range(50):
            v = rounds[k][r].copy()
            L.check(core.lbfgsx_comm_allreduce_sum(comm, r, v.ctypes.data_as(C.POINTER(C.c_double)), 31))
            got[r][k] = v
    th = [threading.Thread(target=worker, args=(r,)) for r in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    for k in range(50):
        want = (rounds[k][0] + rounds[k][1]) + rounds[k][2]       # rank order
        for r in range(3):
            assert np.array_equal(got[r][k], want)
    # a rank that waits alone is released by an abort from outside
    res = []
    v = np.zeros(4)
    t = threading.Thread(target=lambda: res.append(core.lbfgsx_comm_allreduce_sum(comm, 1, v.ctypes.data_as(C.POINTER(C.c_double)), 4)))
    t.start()
    import time
    time.sleep(0.2)
    core.lbfgsx_comm_abort(comm)
    t.join(timeout=10)
    assert not t.is_alive() and res == [L.E_RUNTIME]
    core.lbfgsx_comm_destroy(comm)


@pytest.mark.parametrize("obj,n,devices,m,iters", [("rosen", 40000, [0, 0], 6, 25), ("quad", 30004, [0, 0, 0], 5, 20),
                                                   ("rosen", 200000, [0], 10, 15)])
def test_set_devices_row_shards_one_problem_from_one_process(A, oracle, obj, n, devices, m, iters):
    """LBFGSSolver::set_devices: minimize(f, x) on a host x gives every listed device a row block, its own host thread and
    solver, and sums the driver's dots through lbfgsx_comm_allreduce_sum.  Compared with
      * the ORACLE's vector two-loop on the same problem: same iteration and evaluation counts, every iterate within 1e-8
        over these short runs (the Gram-space recursion equals the vector form only up to rounding, which the trajectory
        then amplifies: this is the stated window of the opt-in mode, not the 1e-10 of the parity path);
      * the un-sharded Gram-space run on one device: 1e-9 (a different split of the same sums)."""
    from lbfgspp_amd import _lib as L
    ls = O.LS_MT if obj == "rosen" else O.LS_NW
    a, b = O.quad_problem(n) if obj == "quad" else (None, None)
    x0 = O.rosen_x0(n, 9) if obj == "rosen" else np.zeros(n)
    f = A.ExtendedRosenbrock() if obj == "rosen" else A.DiagQuadratic(a, b)
    par = A.LBFGSParam(m=m, epsilon=0.0, epsilon_rel=0.0, max_iterations=iters)
    s = A.LBFGSSolver(par, linesearch=ls)
    s.set_devices(devices)
    x = x0.copy()
    niter, fx = s.minimize(f, x)
    nfev = s.last.nfev
    s.set_devices([])
    s.set_recursion(L.RECURSION_GRAM_SPACE)
    x1 = x0.copy()
    niter1, fx1 = s.minimize(f, x1)
    assert (niter, nfev) == (niter1, s.last.nfev)
    assert np.abs(x - x1).max() <= 1e-9 and abs(fx - fx1) <= 1e-9 * max(1.0, abs(fx1))
    x_ref, r = oracle.lbfgs(O.F64, ls, O.OBJ_ROSEN if obj == "rosen" else O.OBJ_QUAD, x0,
                            O.lbfgs_params(m=m, epsilon=0, epsilon_rel=0, max_iterations=iters), a=a, b=b)
    assert (niter, nfev) == (r.niter, r.nfev)
    assert np.abs(x - x_ref).max() <= 1e-8 and abs(fx - r.fx) <= 1e-8 * max(1.0, abs(r.fx))


def test_set_devices_refuses_what_it_cannot_shard(A):
    par = A.LBFGSParam(m=4, epsilon=0.0, epsilon_rel=0.0, max_iterations=3)
    s = A.LBFGSSolver(par, linesearch=O.LS_MT)
    s.set_devices([0, 0, 0, 0])
    with pytest.raises(ValueError, match="fewer than 4 rows"):
        s.minimize(A.ExtendedRosenbrock(), np.zeros(8))
    s.set_devices([0, 7])
    with pytest.raises((ValueError, RuntimeError)):
        s.minimize(A.ExtendedRosenbrock(), np.zeros(64))
    sb = A.LBFGSBSolver(A.LBFGSBParam(m=4))
    with pytest.raises(ValueError):
        sb.set_devices([0, 0])


# ---- real peers: run as soon as a box has two GPUs (the 1-GPU test box skips them; the driver's 8-GPU node does not) -------
def _need_two_gpus(A):
    core, _ = A.load()
    ndev = core.lbfgsx_device_count()
    if ndev < 2:
        pytest.skip("needs >= 2 GPUs: RCCL between distinct devices (this box has %d)" % ndev)
    return ndev


def test_multi_gpu_allreduce_over_distinct_devices(A):
    """lbfgsx_comm_create_local over ALL devices of the node (ncclCommInitAll with distinct peers, xGMI), one host thread per
    device: 40 all-reduces of the solver's bundle size in a row; every rank ends with the same bits and the sums are right."""
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    ndev = _need_two_gpus(A)
    comm, info = _comm_local(A, list(range(ndev)))
    assert info[:3] == (ndev, ndev, 1)
    rng = np.random.default_rng(11)
    rounds = [[rng.standard_normal(67) for _ in range(ndev)] for _ in range(40)]
    got = [[None] * 40 for _ in range(ndev)]
    errs = []

    def worker(r):
        try:
            for k in range(40):
                v = rounds[k][r].copy()
                L.check(core.lbfgsx_comm_allreduce_sum(comm, r, v.ctypes.data_as(C.POINTER(C.c_double)), 67))
                got[r][k] = v
        except BaseException as e:  # noqa: B902
            errs.append(e)
            core.lbfgsx_comm_abort(comm)
    th = [threading.Thread(target=worker, args=(r,)) for r in range(ndev)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert not errs, errs
    for k in range(40):
        want = np.sum(rounds[k], axis=0)
        for r in range(ndev):
            assert np.array_equal(got[r][k], got[0][k])                    # identical bits on every rank
            assert np.allclose(got[r][k], want, rtol=1e-14, atol=1e-14)     # RCCL's order of additions is its own
    assert all(core.lbfgsx_comm_calls(comm, r) == 40 for r in range(ndev))
    core.lbfgsx_comm_destroy(comm)


def test_multi_gpu_record_allgather_over_distinct_devices(A):
    """lbfgsx_rccl_allgather_records with one rank per physical GPU: every device ends with all records, in problem-id
    order, bit for bit."""
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    ndev = _need_two_gpus(A)
    count, rec = 1000 * ndev + 7, 40                                        # not a multiple of the device count
    raw = np.random.default_rng(5).integers(0, 256, size=(count, rec), dtype=np.uint8)
    dv = (C.c_int * ndev)(*range(ndev))
    outp = (C.c_void_p * ndev)()
    L.check(core.lbfgsx_rccl_allgather_records(dv, ndev, raw.ctypes.data_as(C.c_void_p), count, rec, outp))
    try:
        for k in range(ndev):
            back = np.zeros_like(raw)
            L.check(core.lbfgsx_device_download(k, outp[k], raw.size, back.ctypes.data_as(C.c_void_p)))
            assert np.array_equal(back, raw), "device %d" % k
    finally:
        for k in range(ndev):
            core.lbfgsx_device_free(k, outp[k])


@pytest.mark.parametrize("obj,n,m,iters", [("rosen", 400000, 6, 25), ("quad", 300004, 10, 20)])
def test_multi_gpu_set_devices_against_the_oracle(A, oracle, obj, n, m, iters):
    """LBFGSSolver::set_devices over two and over all physical GPUs: the row-sharded run (sums through ncclAllReduce over
    xGMI) against the oracle's vector two-loop -- same counts, every coordinate within 1e-8 (the stated window of the opt-in
    Gram-space recursion) -- and against the single-device Gram-space run (1e-9)."""
    from lbfgspp_amd import _lib as L
    ndev = _need_two_gpus(A)
    ls = O.LS_MT if obj == "rosen" else O.LS_NW
    a, b = O.quad_problem(n) if obj == "quad" else (None, None)
    x0 = O.rosen_x0(n, 9) if obj == "rosen" else np.zeros(n)
    f = A.ExtendedRosenbrock() if obj == "rosen" else A.DiagQuadratic(a, b)
    par = A.LBFGSParam(m=m, epsilon=0.0, epsilon_rel=0.0, max_iterations=iters)
    x_ref, r = oracle.lbfgs(O.F64, ls, O.OBJ_ROSEN if obj == "rosen" else O.OBJ_QUAD, x0,
                            O.lbfgs_params(m=m, epsilon=0, epsilon_rel=0, max_iterations=iters), a=a, b=b)
    s = A.LBFGSSolver(par, linesearch=ls)
    s.set_recursion(L.RECURSION_GRAM_SPACE)
    x1 = x0.copy()
    niter1, fx1 = s.minimize(f, x1)
    for devs in sorted({(0, 1), tuple(range(ndev))}):
        s.set_devices(list(devs))
        x = x0.copy()
        niter, fx = s.minimize(f, x)
        assert (niter, s.last.nfev) == (r.niter, r.nfev) == (niter1, s.last.nfev)
        assert np.abs(x - x1).max() <= 1e-9 and np.abs(x - x_ref).max() <= 1e-8
        assert abs(fx - r.fx) <= 1e-8 * max(1.0, abs(r.fx))
    s.set_devices([])


def test_abort_while_ranks_reduce_is_safe(A):
    """lbfgsx_comm_abort used to free a rank's RCCL communicator while that rank's thread could be between
    its `aborted` check and ncclAllReduce.  The communicator is now handed to RCCL and taken away under the rank's lock: ranks
    that reduce in a loop while another thread aborts must all come back with LBFGSX_E_RUNTIME (or success for calls that
    completed before), never crash; lbfgsx_comm_first_abort names the rank that reported its failure."""
    from lbfgspp_amd import _lib as L
    core, _ = A.load()
    core.lbfgsx_comm_abort_from.restype, core.lbfgsx_comm_abort_from.argtypes = C.c_int, [C.c_void_p, C.c_int]
    core.lbfgsx_comm_first_abort.restype, core.lbfgsx_comm_first_abort.argtypes = C.c_int, [C.c_void_p]
    ndev = core.lbfgsx_device_count()
    for devices in ([0], list(range(ndev)) if ndev > 1 else [0, 0]):
        comm, info = _comm_local(A, devices)
        assert core.lbfgsx_comm_first_abort(comm) == -1
        world = len(devices)
        codes = [[] for _ in range(world)]
        stop = threading.Event()

        def worker(r):
            v = np.ones(16)
            while not stop.is_set():
                rc = core.lbfgsx_comm_allreduce_sum(comm, r, v.ctypes.data_as(C.POINTER(C.c_double)), 16)
                codes[r].append(rc)
                if rc != 0:
                    break
                v[:] = 1.0
        th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
        [t.start() for t in th]
        import time
        time.sleep(0.05)
        core.lbfgsx_comm_abort_from(comm, world - 1)
        stop.set()
        [t.join(30) for t in th]
        assert not any(t.is_alive() for t in th)
        for r in range(world):
            assert all(c == 0 for c in codes[r][:-1]) and codes[r][-1] in (0, L.E_RUNTIME)
        assert core.lbfgsx_comm_first_abort(comm) == world - 1
        v = np.ones(4)
        assert core.lbfgsx_comm_allreduce_sum(comm, 0, v.ctypes.data_as(C.POINTER(C.c_double)), 4) == L.E_RUNTIME
        core.lbfgsx_comm_destroy(comm)
