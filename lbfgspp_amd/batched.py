"""Batched-problems mode (BASELINE.json cfg5): many independent minimisations, sharded over GPUs.

One process per GPU.  Rank r of W solves the contiguous block of problem ids returned by `shard_range`; the
hot path has no inter-GPU traffic at all.  The only collective is the final gather of the per-problem
{niter, nfev, status, fx, gnorm} records (about 32 B per problem), done with torch.distributed -- backend
"nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import ctypes as C

import numpy as np

from . import _lib as L

RECORD = np.dtype([("niter", np.int32), ("nfev", np.int32), ("status", np.int32), ("fx", np.float64),
                   ("gnorm", np.float64)])


def shard_range(nproblems, rank, world):
    """Contiguous, balanced block [first, first+count) of problem ids for `rank` (remainder to the low ranks)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(nproblems, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def solve_local(param, objective, n, first, count, seed_base=1000, algo=L.ALGO_LBFGS, linesearch=L.LS_MORE_THUENTE,
                dtype=np.float32, device=0, nthreads=16):
    """Solve problems [first, first+count) on one GPU; returns a RECORD array of length count."""
    _, sol = L.load()
    items = (L.BatchItem * max(count, 1))()
    cp = param._c()
    dt = L.F64 if np.dtype(dtype) == np.float64 else L.F32
    rc = sol.lbfgsx_batch_minimize(algo, dt, linesearch, C.byref(cp), objective, n, first, count, seed_base, device,
                                   nthreads, items)
    L.check(rc, "lbfgsx_batch_minimize failed (worker could not create its solver)")
    out = np.zeros(count, dtype=RECORD)
    for k in range(count):
        out[k] = (items[k].niter, items[k].nfev, items[k].status, items[k].fx, items[k].gnorm)
    return out


def solve_local_lockstep(param, n, first, count, seed_base=1000, dtype=np.float32, device=0, return_x=False, devices=None,
                         linesearch=L.LS_MORE_THUENTE, objective=L.OBJ_EXT_ROSENBROCK, kappa=10.0):
    """Lock-step batch on one GPU: all `count` problems resident, one launch per statement for the whole batch.
    L-BFGS with LineSearchMoreThuente (default) or LineSearchNocedalWright; built-in objective: extended Rosenbrock
    (default) or the diagonal quadratic of condition number `kappa`, problem id -> data of seed seed_base + id.
    Returns RECORD array [, final iterates (count, n)].
    devices = [d0, d1, ...]: the single-process multi-GPU form: contiguous blocks of problem ids, one per listed device,
    each driven by its own host thread."""
    _, sol = L.load()
    items = (L.BatchItem * max(count, 1))()
    cp = param._c()
    dt = L.F64 if np.dtype(dtype) == np.float64 else L.F32
    xs = np.empty((count, n), dtype=dtype) if return_x else None
    err = C.create_string_buffer(256)
    devs = [int(d) for d in devices] if devices is not None else [int(device)]
    dv = (C.c_int * max(len(devs), 1))(*devs)
    rc = sol.lbfgsx_batch_minimize_lockstep_ex(dt, int(linesearch), int(objective), float(kappa), C.byref(cp), n, first, count,
                                               seed_base, dv, len(devs), items,
                                               xs.ctypes.data_as(C.c_void_p) if return_x else None, err, 256)
    L.check(rc, err.value.decode())
    out = np.zeros(count, dtype=RECORD)
    for k in range(count):
        out[k] = (items[k].niter, items[k].nfev, items[k].status, items[k].fx, items[k].gnorm)
    return (out, xs) if return_x else out


class LockstepBatch:
    """A lock-step batch kept resident across minimisations (lbfgsx_lockstep_create): `count` problems of dimension n on one
    GPU.  minimize() solves the problems of ids [first, first + count) from their start points and returns the RECORD array
    [, the final iterates]; `stats` then holds what that minimisation did (lock-step iterations, launches, host waits and --
    with timing=True -- the sum of the kernels' durations)."""

    def __init__(self, param, n, count, dtype=np.float32, device=0, linesearch=L.LS_MORE_THUENTE, timing=False):
        _, self._sol = L.load()
        self.n, self.count, self.dtype = int(n), int(count), np.dtype(dtype)
        self._h = C.c_void_p()
        self._items = (L.BatchItem * max(self.count, 1))()
        self.stats = None
        err = C.create_string_buffer(256)
        cp = param._c()
        dt = L.F64 if self.dtype == np.float64 else L.F32
        rc = self._sol.lbfgsx_lockstep_create(C.byref(self._h), dt, int(linesearch), C.byref(cp), self.n, self.count,
                                              int(device), 1 if timing else 0, err, 256)
        L.check(rc, err.value.decode())

    def minimize(self, first=0, seed_base=1000, objective=L.OBJ_EXT_ROSENBROCK, kappa=10.0, return_x=False):
        xs = np.empty((self.count, self.n), dtype=self.dtype) if return_x else None
        st = (C.c_double * 8)()
        err = C.create_string_buffer(256)
        rc = self._sol.lbfgsx_lockstep_minimize(self._h, int(objective), float(kappa), int(seed_base), int(first), self._items,
                                                xs.ctypes.data_as(C.c_void_p) if return_x else None, C.byref(st), err, 256)
        L.check(rc, err.value.decode())
        self.stats = {"lockstep_iterations": int(st[0]), "fused": bool(st[1]), "kernel_ms": float(st[2]),
                      "launches": int(st[3]), "waits": int(st[4]), "wait_timeouts": int(st[5])}
        out = np.zeros(self.count, dtype=RECORD)
        for k in range(self.count):
            it = self._items[k]
            out[k] = (it.niter, it.nfev, it.status, it.fx, it.gnorm)
        return (out, xs) if return_x else out

    def set_timing(self, on):
        L.check(self._sol.lbfgsx_lockstep_set_timing(self._h, 1 if on else 0))

    def close(self):
        if self._h:
            self._sol.lbfgsx_lockstep_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gather_records(local, nproblems, rank, world, dist=None, device=None):
    """All ranks obtain the full RECORD array in problem-id order.  `dist` is torch.distributed (initialised)."""
    if world == 1 or dist is None:
        return local
    import torch
    base = -(-nproblems // world)  # padded block length
    buf = np.zeros((base, 5), dtype=np.float64)
    for j, name in enumerate(RECORD.names):
        buf[:len(local), j] = local[name]
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    out = np.zeros(nproblems, dtype=RECORD)
    for r in range(world):
        first, count = shard_range(nproblems, r, world)
        arr = parts[r].cpu().numpy()
        for j, name in enumerate(RECORD.names):
            out[name][first:first + count] = arr[:count, j]
    return out
