"""CPU suite, part 2: the C-ABI libraries load, export every declared symbol, validate parameters like the
reference, and fail loudly (no CPU fallback) when no GPU is present.  No compute calls are made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lbfgsx_[a-z0-9_A-Z]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def libs():
    import lbfgspp_amd as A
    return A.load()


def test_core_library_exports_every_declared_symbol(libs):
    core, _ = libs
    names = _declared("lbfgsx.h")
    assert len(names) > 25
    for n in names:
        assert hasattr(core, n), "liblbfgsx.so does not export %s" % n


def test_solver_library_exports_every_declared_symbol(libs):
    _, sol = libs
    names = [n for n in _declared("lbfgsx_solver.h")]
    assert "lbfgsx_solver_minimize" in names
    for n in names:
        assert hasattr(sol, n), "liblbfgsx_solver.so does not export %s" % n


def test_version_and_device_count(libs):
    core, _ = libs
    assert b"gfx950" in core.lbfgsx_version()
    assert core.lbfgsx_device_count() >= 0


def test_null_context_is_an_invalid_argument_not_a_crash(libs):
    """the Gram-space entries check their context before anything dereferences it."""
    from lbfgspp_amd import _lib as L
    core, _ = libs
    scal = (C.c_double * 7)()
    assert core.lbfgsx_gs_set_history_dtype(None, L.F64) == L.E_INVALID
    assert core.lbfgsx_gs_post_linesearch(None, scal, scal, scal, scal) == L.E_INVALID
    assert core.lbfgsx_gs_direction(None, None, 0.0, None) == L.E_INVALID
    assert core.lbfgsx_device(None) == -1
    out = (C.c_int64 * 3)()
    assert core.lbfgsx_counters(C.byref(out), 0) == 0 and all(v >= 0 for v in out)   # process-wide, needs no context
    ex = (C.c_int64 * 8)()
    assert core.lbfgsx_counters_ex(C.byref(ex), 1) == 0 and list(ex)[:3] == list(out) and all(v >= 0 for v in ex)
    assert core.lbfgsx_counters_ex(C.byref(ex), 0) == 0 and list(ex) == [0] * 8          # reset by the call before
    out4 = (C.c_int64 * 4)()
    assert core.lbfgsx_b_compact_vec_counts(C.byref(out4), 0) == 0 and all(v >= 0 for v in out4)  # process-wide as well
    out2 = (C.c_int64 * 2)()
    assert core.lbfgsx_poll_counts(None, C.byref(out2)) == L.E_INVALID


def test_param_validation_matches_reference_messages():
    """reference Param.h:191-218 / 350-376: same exception type (invalid_argument -> ValueError) and text."""
    import lbfgspp_amd as A
    bad = [(dict(m=0), "'m' must be positive"), (dict(epsilon=-1.0), "'epsilon' must be non-negative"),
           (dict(epsilon_rel=-1.0), "'epsilon_rel' must be non-negative"), (dict(past=-1), "'past' must be non-negative"),
           (dict(delta=-1.0), "'delta' must be non-negative"),
           (dict(max_iterations=-1), "'max_iterations' must be non-negative"),
           (dict(linesearch=0), "unsupported line search termination condition"),
           (dict(max_linesearch=0), "'max_linesearch' must be positive"),
           (dict(min_step=-1.0), "'min_step' must be positive"),
           (dict(max_step=1e-30), "'max_step' must be greater than 'min_step'"),
           (dict(ftol=0.6), "'ftol' must satisfy 0 < ftol < 0.5"),
           (dict(wolfe=1e-5), "'wolfe' must satisfy ftol < wolfe < 1")]
    for kw, msg in bad:
        with pytest.raises(ValueError) as e:
            A.LBFGSSolver(A.LBFGSParam(**kw))
        assert str(e.value) == msg


def test_defaults_match_reference():
    import lbfgspp_amd as A
    p = A.LBFGSParam()
    assert (p.m, p.epsilon, p.epsilon_rel, p.past, p.delta, p.max_iterations, p.linesearch, p.max_linesearch,
            p.min_step, p.max_step, p.ftol, p.wolfe) == (6, 1e-5, 1e-5, 0, 0.0, 0, 3, 20, 1e-20, 1e20, 1e-4, 0.9)
    q = A.LBFGSBParam()
    assert (q.past, q.delta, q.max_submin) == (1, 1e-10, 10)


def test_no_silent_cpu_fallback(libs):
    """Without a GPU every compute entry point must raise; with one, this test is a no-op."""
    core, _ = libs
    if core.lbfgsx_device_count() > 0:
        pytest.skip("GPU present")
    import lbfgspp_amd as A
    s = A.LBFGSSolver(A.LBFGSParam())
    with pytest.raises(RuntimeError, match="no HIP device"):
        s.minimize(A.ExtendedRosenbrock(), np.zeros(10))
    h = C.c_void_p()
    assert core.lbfgsx_create(C.byref(h), 0, 16, 3, 0, 0) == -5  # LBFGSX_E_NOGPU


def test_product_never_imports_the_oracle():
    """The product tree (lbfgspp_amd/, include/) must not load, link or include anything under oracle/."""
    banned = ("oracle_lib", "liboracle", "libref_", "oracle/", "oracle_api", "oracle_port", "oracle_ref", "eigen_shim")
    for base in ("lbfgspp_amd", "include"):
        for root, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".cuh", ".hip", ".cpp")):
                    txt = open(os.path.join(root, f)).read()
                    for b in banned:
                        assert b not in txt, "%s references %s" % (os.path.join(root, f), b)


def test_bench_and_smoke_fail_loudly_without_a_gpu():
    """bench.py / smoke() measure the HIP path only: on a box without a GPU they stop with a clear message instead of
    timing some host path (run here on the CPU container; on the GPU box the condition is simply not met)."""
    import subprocess
    import sys
    core, _ = __import__("lbfgspp_amd").load()
    if core.lbfgsx_device_count() > 0:
        pytest.skip("a GPU is visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode != 0 and "needs a GPU" in r.stdout and '"metric"' not in r.stdout
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode != 0 and "smoke() needs a GPU" in r.stdout


REF_EXAMPLES = "/root/reference/examples"


@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLES), reason="the reference checkout only exists in the build container")
@pytest.mark.parametrize("example", ["example-quadratic.cpp", "example-rosenbrock.cpp", "example-rosenbrock-box.cpp",
                                     "example-rosenbrock-bracketing.cpp", "example-rosenbrock-comparison.cpp"])
def test_reference_examples_build_verbatim_against_the_dropin_headers(tmp_path, example):
    """The reference's example programs, UNMODIFIED and read where they lie, compile and link against include/ +
    liblbfgsx.so with a plain g++ (oracle/eigen_shim stands in for Eigen, which is not installed): Eigen vector types
    through minimize(), final_grad().transpose(), streaming final_approx_hessian(), every line-search policy as the
    template argument, LBFGSBSolver with Eigen bounds.  They run on the GPU box as tests/cpp/test_dropin.cpp (re-typed)
    and tests/cpp/test_reference_policy.cpp (Eigen-typed); here the point is that nothing had to be edited."""
    import subprocess
    lib = os.path.join(ROOT, "lbfgspp_amd")
    if not os.path.exists(os.path.join(lib, "liblbfgsx.so")):
        pytest.skip("liblbfgsx.so not built")
    cmd = ["g++", "-std=c++17", "-O0", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle", "eigen_shim"),
           os.path.join(REF_EXAMPLES, example), "-o", str(tmp_path / "a.out"), "-L" + lib, "-llbfgsx", "-L/opt/rocm/lib",
           "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]


def test_user_policy_program_compiles_without_a_gpu(tmp_path):
    import subprocess
    cmd = ["g++", "-std=c++17", "-O0", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "oracle", "eigen_shim"), os.path.join(ROOT, "tests", "cpp", "test_reference_policy.cpp")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    # and the std::vector flavour still builds with Eigen hidden
    cmd = ["g++", "-std=c++17", "-O0", "-fsyntax-only", "-DLBFGSX_NO_EIGEN", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "oracle", "eigen_shim"), os.path.join(ROOT, "tests", "cpp", "test_dropin.cpp")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
