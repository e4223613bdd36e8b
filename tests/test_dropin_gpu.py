"""-m gpu: compile a plain C++ user program against include/ + liblbfgsx.so (g++ only, no hipcc) and run it:
host functors with the reference's signature, the reference's example problems, a device functor."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_dropin_program(tmp_path):
    exe = str(tmp_path / "test_dropin")
    lib = os.path.join(ROOT, "lbfgspp_amd")
    # the HIP runtime is whatever liblbfgsx.so itself resolves (hipMemcpy is taken from there)
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_dropin.cpp"),
           "-o", exe, "-L" + lib, "-llbfgsx", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0 and "DROPIN OK" in out.stdout, out.stdout


def test_reference_style_program_with_user_policy(tmp_path):
    """A program in the reference's own types -- Eigen vectors (the stand-in under oracle/eigen_shim), getters used as
    the reference's examples use them, and a user line-search policy with the reference's ten-argument signature
    (LBFGS.h:20-21,127) -- builds against include/ and runs: same decisions as the built-in Armijo policy."""
    exe = str(tmp_path / "test_reference_policy")
    lib = os.path.join(ROOT, "lbfgspp_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "oracle", "eigen_shim"), os.path.join(ROOT, "tests", "cpp", "test_reference_policy.cpp"),
           "-o", exe, "-L" + lib, "-llbfgsx", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0 and "POLICY OK" in out.stdout, out.stdout


@pytest.mark.parametrize("n,m,npairs,dtype", [(1000, 6, 3, "f64"), (4099, 5, 13, "f64"), (40, 4, 7, "f64"), (65537, 10, 10, "f32"),
                                             (33, 3, 0, "f32")])
def test_reference_class_name_drives_the_matrix_directly(tmp_path, n, m, npairs, dtype):
    """`LBFGSpp::BFGSMat<Scalar>` with the reference's member signatures over host vectors (BFGSMat.h:61,81,276,307,310;
    the dense getters :150,:211): a program that calls reset / add_correction / apply_Hv itself builds with g++ alone and
    gets the product of the reference's own class (oracle/_ref) on the same pairs, the ring wrapped when npairs > m."""
    import numpy as np

    import oracle_lib as O

    if not O.available("ref"):
        pytest.skip("oracle/_ref not built")
    exe = str(tmp_path / "test_bfgsmat_class")
    lib = os.path.join(ROOT, "lbfgspp_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "oracle", "eigen_shim"), os.path.join(ROOT, "tests", "cpp", "test_bfgsmat_class.cpp"),
           "-o", exe, "-L" + lib, "-llbfgsx", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    path = str(tmp_path / "out.bin")
    out = subprocess.run([exe, str(n), str(m), str(npairs), dtype, path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0 and "BFGSMAT OK" in out.stdout, out.stdout
    dt = np.float64 if dtype == "f64" else np.float32
    raw = open(path, "rb").read()
    body = np.frombuffer(raw[:-16], dt)
    theta, ncorr = np.frombuffer(raw[-16:], np.float64)
    assert body.size == (2 * npairs + 2) * n
    pairs = body[:2 * npairs * n].reshape(npairs, 2, n)
    S, Y = np.ascontiguousarray(pairs[:, 0]), np.ascontiguousarray(pairs[:, 1])
    v, got = body[2 * npairs * n:(2 * npairs + 1) * n], body[(2 * npairs + 1) * n:]
    assert int(ncorr) == min(npairs, m)
    ref = O.Oracle("ref").apply_Hv(O.F64 if dtype == "f64" else O.F32, m, S, Y, v.copy(), -1.0)
    scale = np.abs(ref).max() + 1e-300
    assert np.abs(got - ref).max() <= 4 * np.finfo(dt).eps * scale
    assert np.mean(got == ref) > 0.99
    if npairs:
        sy = float(np.dot(S[-1].astype(np.float64), Y[-1].astype(np.float64)))
        yy = float(np.dot(Y[-1].astype(np.float64), Y[-1].astype(np.float64)))
        assert abs(theta - yy / sy) <= (1e-5 if dtype == "f32" else 1e-12) * abs(yy / sy)
    else:
        assert theta == 1.0
