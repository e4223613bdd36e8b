"""Python mirror of the reference's solver API over the native MI355X path.

Names, argument meaning and error behaviour follow yixuan/LBFGSpp:
  LBFGSParam / LBFGSBParam   -- reference include/LBFGSpp/Param.h:67-219, 224-377 (same fields and defaults)
  LBFGSSolver(param, linesearch).minimize(f, x)  -> (niter, fx)      reference include/LBFGS.h:78-173
  LBFGSBSolver(param).minimize(f, x, lb, ub)     -> (niter, fx)      reference include/LBFGSB.h:116-262
Exceptions: std::invalid_argument -> ValueError, std::logic_error -> ArithmeticError,
std::runtime_error -> RuntimeError (same messages).  `f` is a built-in device objective
(`DiagQuadratic(a, b)`, `ExtendedRosenbrock()`); all O(n) work runs in the HIP library, Python only passes
pointers.  There is no CPU fallback: without the built extension or a GPU these calls raise.
"""
import ctypes as C

import numpy as np

from . import _lib as L

_NP = {L.F64: np.float64, L.F32: np.float32}


class LBFGSParam:
    """Reference Param.h:168-184 defaults."""

    def __init__(self, **kw):
        self.m = 6
        self.epsilon = 1e-5
        self.epsilon_rel = 1e-5
        self.past = 0
        self.delta = 0.0
        self.max_iterations = 0
        self.linesearch = 3  # LBFGS_LINESEARCH_BACKTRACKING_STRONG_WOLFE
        self.max_linesearch = 20
        self.min_step = 1e-20
        self.max_step = 1e20
        self.ftol = 1e-4
        self.wolfe = 0.9
        self.max_submin = 10
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)

    def _c(self):
        return L.Params(m=self.m, epsilon=self.epsilon, epsilon_rel=self.epsilon_rel, past=self.past,
                        delta=self.delta, max_iterations=self.max_iterations, linesearch=self.linesearch,
                        max_linesearch=self.max_linesearch, min_step=self.min_step, max_step=self.max_step,
                        ftol=self.ftol, wolfe=self.wolfe, max_submin=self.max_submin)


class LBFGSBParam(LBFGSParam):
    """Reference Param.h:327-343 defaults (past = 1, delta = 1e-10, max_submin = 10)."""

    def __init__(self, **kw):
        super().__init__(**{**dict(past=1, delta=1e-10), **kw})


class DiagQuadratic:
    """f(x) = 0.5*||a.*x - b||^2 ; a, b host arrays, or None when generated on the device."""
    objective = L.OBJ_DIAG_QUAD

    def __init__(self, a=None, b=None):
        self.a, self.b = a, b


class ExtendedRosenbrock:
    """sum over pairs (1-x0)^2 + 100 (x1-x0^2)^2 in the reference's example form."""
    objective = L.OBJ_EXT_ROSENBROCK
    a = b = None


class TraceBuffer:
    """Per-evaluation record (fx and x[::stride]) for the parity tests."""

    def __init__(self, n, cap=512, stride=1, with_x=True):
        self.nsamp = (n + stride - 1) // stride
        self.fx = np.zeros(cap, dtype=np.float64)
        self.xs = np.zeros((cap, self.nsamp), dtype=np.float64) if with_x else None
        pd = C.POINTER(C.c_double)
        self.c = L.Trace(cap=cap, count=0, fx=self.fx.ctypes.data_as(pd), stride=stride, nsamp=self.nsamp,
                         xs=self.xs.ctypes.data_as(pd) if with_x else None)

    @property
    def count(self):
        return self.c.count


class Result:
    def __init__(self, r):
        self.niter, self.nfev, self.fx, self.gnorm = r.niter, r.nfev, r.fx, r.gnorm
        self.status, self.msg = r.status, r.msg.decode()


class _SolverBase:
    _algo = L.ALGO_LBFGS

    def __init__(self, param, linesearch=L.LS_NOCEDAL_WRIGHT, dtype=np.float64, device=0):
        self._core, self._sol = L.load()
        self.dtype = L.F64 if np.dtype(dtype) == np.float64 else L.F32
        self.param = param
        self._h = C.c_void_p()
        cp = param._c()
        rc = self._sol.lbfgsx_solver_create(C.byref(self._h), self._algo, self.dtype, linesearch, C.byref(cp), device)
        L.check(rc, self._sol.lbfgsx_solver_create_error().decode())
        self.last = None

    def __del__(self):
        try:
            if self._h:
                self._sol.lbfgsx_solver_destroy(self._h)
                self._h = None
        except Exception:
            pass

    close = __del__

    def prepare(self, n):
        """Allocate the device state for dimension n; returns the low-level context handle."""
        rc = self._sol.lbfgsx_solver_prepare(self._h, n)
        L.check(rc)
        return C.c_void_p(self._sol.lbfgsx_solver_ctx(self._h))

    def set_iteration_hook(self, fn):
        """fn(k) is called on the host after iteration k produced the next search direction (None to clear)."""
        self._hook = L.ITER_HOOK((lambda k, _u: fn(k)) if fn else 0)
        L.check(self._sol.lbfgsx_solver_set_iteration_hook(self._h, self._hook, None))

    def set_recursion(self, form):
        """Extension: L.RECURSION_VECTOR (bit-parity two-loop, default) or L.RECURSION_GRAM_SPACE (coefficient-space
        recursion over [S, Y, g]: about half the HBM traffic, equal to the vector form only up to rounding)."""
        L.check(self._sol.lbfgsx_solver_set_recursion(self._h, int(form)), "set_recursion: unknown form, or not an L-BFGS solver")

    def set_reducer(self, fn):
        """Extension, row-sharded runs: fn(values) receives a float64 numpy view of a small array and must replace it by
        its sum over all ranks (an all-reduce).  None switches back.  Gram-space recursion only."""
        if fn is None:
            self._red = L.ALLREDUCE(0)
        else:
            def cb(ptr, count, _user):
                fn(np.ctypeslib.as_array(ptr, shape=(count,)))
            self._red = L.ALLREDUCE(cb)
        L.check(self._sol.lbfgsx_solver_set_allreduce(self._h, self._red, None), "set_reducer: not an L-BFGS solver")

    def set_devices(self, devices):
        """Extension: minimize(f, x) row-shards the ONE problem over these GPUs of the node from this process (one host
        thread + context per device; the driver's sums cross the devices through the library's RCCL all-reduce).
        [] switches back."""
        devs = [int(d) for d in (devices or [])]
        arr = (C.c_int * max(len(devs), 1))(*devs)
        L.check(self._sol.lbfgsx_solver_set_devices(self._h, arr, len(devs)), "set_devices: not an L-BFGS solver")

    def set_native_reducer(self, comm, local_rank=0):
        """Extension, row-sharded runs: the reducer is the library's own all-reduce over `comm` (lbfgsx_comm_create_*),
        called from C without passing through Python."""
        core, _ = L.load()
        hook = C.cast(core.lbfgsx_comm_allreduce_hook, L.ALLREDUCE)
        L.check(self._sol.lbfgsx_solver_set_allreduce(self._h, hook, C.c_void_p(core.lbfgsx_comm_hook_arg(comm, local_rank))),
                "set_reducer: not an L-BFGS solver")

    @property
    def ctx(self):
        return C.c_void_p(self._sol.lbfgsx_solver_ctx(self._h))

    def _ptr(self, arr):
        if arr is None:
            return None
        assert arr.dtype == _NP[self.dtype] and arr.flags["C_CONTIGUOUS"]
        return arr.ctypes.data_as(C.c_void_p)

    def _minimize(self, f, n, x, lb, ub, trace):
        res = L.Result()
        a = None if f.a is None else np.ascontiguousarray(f.a, _NP[self.dtype])
        b = None if f.b is None else np.ascontiguousarray(f.b, _NP[self.dtype])
        rc = self._sol.lbfgsx_solver_minimize(self._h, f.objective, n, self._ptr(a), self._ptr(b), self._ptr(x),
                                              self._ptr(lb), self._ptr(ub), C.byref(trace.c) if trace else None,
                                              C.byref(res))
        self.last = Result(res)
        L.check(rc, self.last.msg)
        return self.last


class LBFGSSolver(_SolverBase):
    """LBFGSSolver<Scalar, LineSearch> (reference LBFGS.h:20-23)."""
    _algo = L.ALGO_LBFGS

    def minimize(self, f, x, trace=None):
        """x: numpy vector, updated in place.  Returns (niter, fx) like minimize(f, x, fx)."""
        xx = np.ascontiguousarray(x, _NP[self.dtype])
        r = self._minimize(f, xx.size, xx, None, None, trace)
        if xx is not x:
            x[...] = xx
        return r.niter, r.fx

    def minimize_resident(self, f, n, trace=None):
        """Start point already in LBFGSX_VEC_X of the device state (see prepare()); result stays there."""
        r = self._minimize(f, n, None, None, None, trace)
        return r.niter, r.fx

    def final_grad_norm(self):
        return self.last.gnorm

    def final_approx_hessians(self, n):
        """(final_approx_hessian(), final_approx_inverse_hessian()) as dense n x n arrays (small n only)."""
        B = np.zeros((n, n), order="F")
        H = np.zeros((n, n), order="F")
        L.check(self._sol.lbfgsx_solver_hessians(self._h, B.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p)))
        return B, H


class LBFGSBSolver(_SolverBase):
    """LBFGSBSolver<Scalar> with LineSearchMoreThuente (reference LBFGSB.h:21-23)."""
    _algo = L.ALGO_LBFGSB

    def __init__(self, param, dtype=np.float64, device=0):
        super().__init__(param, linesearch=L.LS_MORE_THUENTE, dtype=dtype, device=device)

    def minimize(self, f, x, lb, ub, trace=None):
        """minimize(f, x, fx, lb, ub): x updated in place; raises ValueError when lb/ub sizes differ from x."""
        dt = _NP[self.dtype]
        xx = np.ascontiguousarray(x, dt)
        lbv, ubv = np.ascontiguousarray(lb, dt), np.ascontiguousarray(ub, dt)
        if lbv.size != xx.size or ubv.size != xx.size:
            raise ValueError("'lb' and 'ub' must have the same size as 'x'")
        r = self._minimize(f, xx.size, xx, lbv, ubv, trace)
        if xx is not x:
            x[...] = xx
        return r.niter, r.fx

    def minimize_resident(self, f, n, trace=None):
        """x0, lb, ub already in VEC_X / VEC_LB / VEC_UB of the device state (see prepare())."""
        r = self._minimize(f, n, None, None, None, trace)
        return r.niter, r.fx

    def final_grad_norm(self):
        return self.last.gnorm

    def stats(self):
        arr = (C.c_longlong * 8)()
        L.check(self._sol.lbfgsx_solver_stats(self._h, C.byref(arr)))
        keys = ("gcp_crossings", "submin_sweeps", "submin_calls", "submin_unconverged", "resets", "gcp_build_us",
                "gcp_fetch_us", "gcp_total_us")
        d = dict(zip(keys, list(arr)))
        arr2 = (C.c_longlong * 8)()
        L.check(self._sol.lbfgsx_solver_stats2(self._h, C.byref(arr2)))
        d.update(zip(("gcp_dev_crossings", "gcp_sort_fallbacks", "gcp_partial_sorts", "submin_us", "linesearch_us",
                      "correction_us", "submin_fused_sweeps", "gram_carried"), list(arr2)[:8]))
        arr3 = (C.c_longlong * 8)()
        L.check(self._sol.lbfgsx_solver_stats3(self._h, C.byref(arr3)))
        d.update(zip(("gcp_searches", "gcp_nord", "gcp_sorted", "rhs_identities"), list(arr3)[:4]))
        return d
