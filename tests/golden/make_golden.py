#!/usr/bin/env python
"""Generate tests/golden/*.json from the reference itself (oracle/_ref = unmodified reference headers +
oracle/eigen_shim, double-double accumulators).  Run in the build container where /root/reference exists:

    make -C oracle ref && python tests/golden/make_golden.py

The fixtures pin (a) the known answers quoted in SURVEY.md 8(c) / the reference's README and examples and
(b) seeded trajectories (per-evaluation fx, sampled x) that the restatement (oracle/lbfgs_oracle.cpp) and the
HIP path must reproduce.  Floats are stored as hex strings (bit-exact)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def hx(a):
    return [float(v).hex() for v in np.asarray(a, dtype=np.float64).ravel()]


def lbfgs_case(orc, name, dtype, ls, obj, n, m, iters, seed=7, kappa=10.0, stride=None, **pk):
    if obj == O.OBJ_ROSEN:
        x0 = O.rosen_x0(n, seed, dtype) if pk.pop("hash_x0", True) else np.zeros(n, O.NPDT[dtype])
        a = b = None
    else:
        x0 = np.zeros(n, O.NPDT[dtype])
        a, b = O.quad_problem(n, kappa, 1, dtype)
        pk.pop("hash_x0", None)
    p = O.lbfgs_params(m=m, max_iterations=iters, **pk)
    stride = stride or max(1, n // 64)
    tr = O.TraceBuf(n, cap=1024, stride=stride)
    x, r = orc.lbfgs(dtype, ls, obj, x0, p, a=a, b=b, trace=tr)
    k = tr.count
    return dict(name=name, algo="lbfgs", dtype=dtype, ls=ls, obj=obj, n=n, m=m, max_iterations=iters, seed=seed,
                kappa=kappa, params={k_: getattr(p, k_) for k_, _ in p._fields_}, hash_x0=bool(obj == O.OBJ_ROSEN and np.any(x0 != 0)),
                niter=r.niter, nfev=r.nfev, status=r.status, msg=r.msg.decode(), fx=float(r.fx).hex(),
                gnorm=float(r.gnorm).hex(), stride=stride, trace_fx=hx(tr.fx[:k]), trace_xs=hx(tr.xs[:k]),
                x_sample=hx(x[::stride]))


def lbfgsb_instance(seed, n, npairs, mode):
    """Deterministic Cauchy/subspace test instance (same generator as tests/test_lbfgsb_gpu.py)."""
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


def make_lbfgsb(orc):
    out = dict(generator="tests/golden/make_golden.py", oracle=orc.description, trajectories=[], instances=[])
    for name, n, m, iters, stride in (("boxquad_n2000_m6", 2000, 6, 15, 4), ("boxquad_n6000_m10", 6000, 10, 20, 16)):
        a, b = O.quad_problem(n)
        p = O.lbfgsb_params(m=m, epsilon=0.0, epsilon_rel=0.0, past=0, max_iterations=iters)
        tr = O.TraceBuf(n, cap=1024, stride=stride)
        x, r = orc.lbfgsb(O.F64, O.OBJ_QUAD, np.zeros(n), -np.ones(n), np.ones(n), p, a=a, b=b, trace=tr)
        k = tr.count
        out["trajectories"].append(dict(name=name, n=n, m=m, max_iterations=iters, stride=stride, niter=r.niter,
                                        nfev=r.nfev, fx=float(r.fx).hex(), gnorm=float(r.gnorm).hex(),
                                        trace_fx=hx(tr.fx[:k]), trace_xs=hx(tr.xs[:k]), x_sample=hx(x[::stride])))
    for seed, n, m, npairs, mode in ((1, 512, 6, 0, "hard"), (2, 512, 6, 4, "hard"), (3, 768, 5, 9, "edge"),
                                     (4, 640, 8, 8, "gentle")):
        S, Y, x0, g, lb, ub = lbfgsb_instance(seed, n, npairs, mode)
        res = orc.cauchy_subspace(O.F64, m, S, Y, x0, g, lb, ub, max_submin=10)
        out["instances"].append(dict(seed=seed, n=n, m=m, npairs=npairs, mode=mode, xcp=hx(res["xcp"]),
                                     vecc=hx(res["vecc"]), drt=hx(res["drt"]), newact=[int(v) for v in res["newact"]],
                                     fv=[int(v) for v in res["fv"]]))
    with open(os.path.join(HERE, "lbfgsb_golden.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote L-BFGS-B golden:", [(t["name"], t["niter"], t["nfev"]) for t in out["trajectories"]],
          [(i["seed"], len(i["newact"]), len(i["fv"])) for i in out["instances"]])


def make_hessians(orc):
    """final_approx_hessian() / final_approx_inverse_hessian() after minimize() (reference LBFGS.h:192-197,
    BFGSMat.h:150-271): the README example (n=10, x0=0) and two seeded cases with a wrapped history."""
    out = dict(generator="tests/golden/make_golden.py", oracle=orc.description, cases=[])
    for name, ls, obj, n, m, iters, hashed in (("readme_rosen10_nw", O.LS_NW, O.OBJ_ROSEN, 10, 6, 100, False),
                                               ("rosen12_mt_m3_wrapped", O.LS_MT, O.OBJ_ROSEN, 12, 3, 8, True),
                                               ("quad9_nw_m4_short", O.LS_NW, O.OBJ_QUAD, 9, 4, 2, False)):
        if obj == O.OBJ_ROSEN:
            x0 = O.rosen_x0(n, 7, O.F64) if hashed else np.zeros(n)
            a = b = None
        else:
            x0 = np.zeros(n)
            a, b = O.quad_problem(n, 10.0, 1, O.F64)
        p = O.lbfgs_params(m=m, max_iterations=iters, epsilon=1e-6)
        x, r, B, H = orc.lbfgs_hessians(O.F64, ls, obj, x0, p, a=a, b=b)
        out["cases"].append(dict(name=name, ls=ls, obj=obj, n=n, m=m, max_iterations=iters, hash_x0=hashed, niter=r.niter,
                                 B=hx(B.ravel(order="F")), H=hx(H.ravel(order="F"))))
    with open(os.path.join(HERE, "hessian_golden.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote hessian golden:", [(c["name"], c["niter"]) for c in out["cases"]])


def main():
    orc = O.Oracle("ref", "dd")
    make_hessians(orc)
    cases = []
    # README / example known answers: Rosenbrock n=10, x0=0
    for ls, nm in ((O.LS_NW, "nw"), (O.LS_MT, "mt"), (O.LS_BT, "bt"), (O.LS_BR, "br")):
        cases.append(lbfgs_case(orc, "readme_rosen10_f64_" + nm, O.F64, ls, O.OBJ_ROSEN, 10, 6, 100, hash_x0=False,
                                epsilon=1e-6, stride=1))
    cases.append(lbfgs_case(orc, "readme_rosen10_f64_nw_epsrel0", O.F64, O.LS_NW, O.OBJ_ROSEN, 10, 6, 100,
                            hash_x0=False, epsilon=1e-6, epsilon_rel=0.0, stride=1))
    cases.append(lbfgs_case(orc, "example_rosen10_f32_nw", O.F32, O.LS_NW, O.OBJ_ROSEN, 10, 6, 0, hash_x0=False, stride=1))
    # seeded trajectories (fixed work: epsilon = epsilon_rel = 0)
    for ls, nm in ((O.LS_NW, "nw"), (O.LS_MT, "mt")):
        cases.append(lbfgs_case(orc, "rosen_n4096_m6_f64_" + nm, O.F64, ls, O.OBJ_ROSEN, 4096, 6, 25, epsilon=0.0, epsilon_rel=0.0))
        cases.append(lbfgs_case(orc, "quad_n5001_m10_f64_" + nm, O.F64, ls, O.OBJ_QUAD, 5001, 10, 25, epsilon=0.0, epsilon_rel=0.0))
        cases.append(lbfgs_case(orc, "rosen_n4098_m5_f32_" + nm, O.F32, ls, O.OBJ_ROSEN, 4098, 5, 12, seed=1000, epsilon=0.0, epsilon_rel=0.0))
    make_lbfgsb(orc)
    with open(os.path.join(HERE, "lbfgs_golden.json"), "w") as f:
        json.dump(dict(generator="tests/golden/make_golden.py", oracle=orc.description, cases=cases), f, indent=0)
    print("wrote", len(cases), "L-BFGS cases")
    for c in cases:
        print(" ", c["name"], c["niter"], c["nfev"], c["status"], float.fromhex(c["fx"]))


if __name__ == "__main__":
    main()
