"""ctypes loader for the CPU oracles (TEST INFRASTRUCTURE ONLY -- never imported by lbfgspp_amd).

Two families export the same C surface (oracle/oracle_api.h):
  * ``ref``  : oracle/_ref/libref_<acc>.so  -- the unmodified reference headers + oracle/eigen_shim
  * ``port`` : oracle/liboracle_<acc>.so    -- the self-contained restatement oracle/lbfgs_oracle.cpp
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

F64, F32 = 0, 1
LS_NW, LS_MT, LS_BT, LS_BR = 0, 1, 2, 3
OBJ_QUAD, OBJ_ROSEN = 0, 1
NPDT = {F64: np.float64, F32: np.float32}


class Params(C.Structure):
    _fields_ = [("m", C.c_int), ("epsilon", C.c_double), ("epsilon_rel", C.c_double), ("past", C.c_int),
                ("delta", C.c_double), ("max_iterations", C.c_int), ("linesearch", C.c_int),
                ("max_linesearch", C.c_int), ("min_step", C.c_double), ("max_step", C.c_double),
                ("ftol", C.c_double), ("wolfe", C.c_double), ("max_submin", C.c_int)]


def lbfgs_params(**kw):
    """LBFGSParam defaults (reference Param.h:168-184)."""
    p = Params(m=6, epsilon=1e-5, epsilon_rel=1e-5, past=0, delta=0.0, max_iterations=0, linesearch=3,
               max_linesearch=20, min_step=1e-20, max_step=1e20, ftol=1e-4, wolfe=0.9, max_submin=10)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def lbfgsb_params(**kw):
    """LBFGSBParam defaults (reference Param.h:327-343)."""
    return lbfgs_params(**{**dict(past=1, delta=1e-10), **kw})


class Result(C.Structure):
    _fields_ = [("niter", C.c_int), ("nfev", C.c_int), ("fx", C.c_double), ("gnorm", C.c_double),
                ("status", C.c_int), ("msg", C.c_char * 200)]


class Trace(C.Structure):
    _fields_ = [("cap", C.c_int), ("count", C.c_int), ("fx", C.POINTER(C.c_double)), ("stride", C.c_long),
                ("nsamp", C.c_long), ("xs", C.POINTER(C.c_double))]


class TraceBuf:
    def __init__(self, n, cap=512, stride=1, with_x=True):
        self.nsamp = (n + stride - 1) // stride
        self.fx = np.zeros(cap, dtype=np.float64)
        self.xs = np.zeros((cap, self.nsamp), dtype=np.float64) if with_x else None
        self.c = Trace(cap=cap, count=0, fx=self.fx.ctypes.data_as(C.POINTER(C.c_double)), stride=stride,
                       nsamp=self.nsamp,
                       xs=self.xs.ctypes.data_as(C.POINTER(C.c_double)) if with_x else None)

    @property
    def count(self):
        return self.c.count


def _path(family, acc):
    if family == "ref":
        return os.path.join(ROOT, "oracle", "_ref", "libref_%s.so" % acc)
    return os.path.join(ROOT, "oracle", "liboracle_%s.so" % acc)


def available(family, acc="dd"):
    return os.path.exists(_path(family, acc))


class Oracle:
    def __init__(self, family="ref", acc="dd"):
        self.family, self.acc = family, acc
        self.lib = C.CDLL(_path(family, acc))
        pre = "oracle_ref" if family == "ref" else "oracle_port"
        self._lbfgs = getattr(self.lib, pre + "_lbfgs")
        self._lbfgsb = getattr(self.lib, pre + "_lbfgsb")
        self._hv = getattr(self.lib, pre + "_apply_Hv")
        self._eval = getattr(self.lib, pre + "_eval")
        self._eval.restype = C.c_double
        self._cs = getattr(self.lib, pre + "_cauchy_subspace")
        self._hess = getattr(self.lib, pre + "_lbfgs_hessians")
        d = getattr(self.lib, pre + "_describe")
        d.restype = C.c_char_p
        self.description = d().decode()
        vp = C.c_void_p
        self._lbfgs.argtypes = [C.c_int, C.c_int, C.c_int, C.c_long, vp, vp, vp, C.POINTER(Params),
                                C.POINTER(Trace), C.POINTER(Result)]
        self._lbfgsb.argtypes = [C.c_int, C.c_int, C.c_long, vp, vp, vp, vp, vp, C.POINTER(Params),
                                 C.POINTER(Trace), C.POINTER(Result)]
        self._hv.argtypes = [C.c_int, C.c_long, C.c_int, C.c_int, vp, vp, vp, C.c_double, vp]
        self._eval.argtypes = [C.c_int, C.c_int, C.c_long, vp, vp, vp, vp]
        self._cs.argtypes = [C.c_int, C.c_long, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp,
                             C.POINTER(C.c_int), vp, C.POINTER(C.c_int), vp]

    def set_replication(self, r):
        """Restatement only (oracle/acc.h): every n-length sum of the L-BFGS path is multiplied by r (a power of two), so
        a run on a base problem of size p is the exact image of the same problem tiled r times.  1 switches it off."""
        if self.family != "port":
            raise RuntimeError("replicated-problem mode exists in the restatement only")
        f = self.lib.oracle_port_set_replication
        f.argtypes = [C.c_double]
        f.restype = C.c_int
        if f(float(r)) != 0:
            raise ValueError("replication factor must be a power of two >= 1")

    @property
    def supports_lbfgsb(self):
        """False for the restatement until its L-BFGS-B part is built (entry points answer -1000)."""
        x = np.zeros(2)
        _, r = self.lbfgsb(F64, OBJ_QUAD, x, -np.ones(2), np.ones(2), lbfgsb_params(max_iterations=1),
                           a=np.ones(2), b=np.ones(2))
        return r.status != -1000

    @staticmethod
    def _p(arr):
        return None if arr is None else arr.ctypes.data_as(C.c_void_p)

    def lbfgs(self, dtype, ls, obj, x0, params, a=None, b=None, trace=None):
        x = np.ascontiguousarray(x0, dtype=NPDT[dtype]).copy()
        res = Result()
        self._lbfgs(dtype, ls, obj, x.size, self._p(a), self._p(b), self._p(x), C.byref(params),
                    C.byref(trace.c) if trace else None, C.byref(res))
        return x, res

    def lbfgsb(self, dtype, obj, x0, lb, ub, params, a=None, b=None, trace=None):
        x = np.ascontiguousarray(x0, dtype=NPDT[dtype]).copy()
        res = Result()
        self._lbfgsb(dtype, obj, x.size, self._p(a), self._p(b), self._p(lb), self._p(ub), self._p(x),
                     C.byref(params), C.byref(trace.c) if trace else None, C.byref(res))
        return x, res

    def lbfgs_hessians(self, dtype, ls, obj, x0, params, a=None, b=None):
        """minimize() then (final_approx_hessian, final_approx_inverse_hessian); None when unsupported."""
        x = np.ascontiguousarray(x0, dtype=NPDT[dtype]).copy()
        n = x.size
        B = np.zeros((n, n), order="F")
        H = np.zeros((n, n), order="F")
        res = Result()
        self._hess.argtypes = [C.c_int, C.c_int, C.c_int, C.c_long] + [C.c_void_p] * 3 + [C.POINTER(Params), C.c_void_p,
                                                                                       C.c_void_p, C.POINTER(Result)]
        rc = self._hess(dtype, ls, obj, n, self._p(a), self._p(b), self._p(x), C.byref(params), self._p(B), self._p(H),
                        C.byref(res))
        if rc == -1000:
            return None
        return x, res, B, H

    def apply_Hv(self, dtype, m, S, Y, v, alpha):
        """S, Y: [npairs, n] arrays (row k = k-th correction pair fed to add_correction)."""
        S = np.ascontiguousarray(S, dtype=NPDT[dtype])
        Y = np.ascontiguousarray(Y, dtype=NPDT[dtype])
        v = np.ascontiguousarray(v, dtype=NPDT[dtype])
        res = np.empty_like(v)
        self._hv(dtype, v.size, m, S.shape[0], self._p(S), self._p(Y), self._p(v), float(alpha), self._p(res))
        return res

    def eval(self, dtype, obj, x, a=None, b=None):
        x = np.ascontiguousarray(x, dtype=NPDT[dtype])
        g = np.empty_like(x)
        fx = self._eval(dtype, obj, x.size, self._p(a), self._p(b), self._p(x), self._p(g))
        return fx, g

    def cauchy_subspace(self, dtype, m, S, Y, x0, g, lb, ub, max_submin=10, subspace=True):
        dt = NPDT[dtype]
        S = np.ascontiguousarray(S, dtype=dt).reshape(-1, x0.size)
        Y = np.ascontiguousarray(Y, dtype=dt).reshape(-1, x0.size)
        n = x0.size
        npairs = S.shape[0]
        ncorr = min(npairs, m)
        xcp = np.empty(n, dt)
        vecc = np.zeros(2 * ncorr, dt)
        newact = np.zeros(n, np.int32)
        fv = np.zeros(n, np.int32)
        nn, nf = C.c_int(0), C.c_int(0)
        drt = np.empty(n, dt) if subspace else None
        self._cs(dtype, n, m, npairs, self._p(S), self._p(Y), self._p(np.ascontiguousarray(x0, dt)),
                 self._p(np.ascontiguousarray(g, dt)), self._p(np.ascontiguousarray(lb, dt)),
                 self._p(np.ascontiguousarray(ub, dt)), max_submin, self._p(xcp), self._p(vecc),
                 self._p(newact), C.byref(nn), self._p(fv), C.byref(nf), self._p(drt))
        return dict(xcp=xcp, vecc=vecc, newact=newact[:nn.value].copy(), fv=fv[:nf.value].copy(), drt=drt)


# ---- synthetic problems; vectorised numpy mirror of oracle/problems.h
_M64 = (1 << 64) - 1


def splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def u01(n, seed):
    with np.errstate(over="ignore"):
        i = np.arange(n, dtype=np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) & _M64)
        return (splitmix64(i) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def quad_problem(n, kappa=10.0, seed=1, dtype=F64):
    i = np.arange(n, dtype=np.float64)
    a = 1.0 + (kappa - 1.0) * (i / float(n - 1)) if n > 1 else np.ones(1)
    b = a * (4.0 * u01(n, seed) - 2.0)
    return a.astype(NPDT[dtype]), b.astype(NPDT[dtype])


def rosen_x0(n, seed=7, dtype=F64):
    base = np.where(np.arange(n) % 2 == 1, 1.0, -1.2)
    return (base + 0.4 * u01(n, seed)).astype(NPDT[dtype])
