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
