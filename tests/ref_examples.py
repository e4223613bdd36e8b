"""TEST INFRASTRUCTURE: the reference's five example programs, compiled UNMODIFIED (read where they lie under
/root/reference/examples, never copied) in two flavours into the git-ignored tests/cpp/bin/:

    <name>.gpu   against include/ (the drop-in headers) + liblbfgsx.so      -- runs on the MI355X
    <name>.ref   against /root/reference/include (the reference's own headers) -- runs on the host

Both see oracle/eigen_shim as <Eigen/Core> (double-double accumulators, the parity build of DESIGN.md section 2), the same
-ffp-contract=off, and tests/cpp/precise_cout.h force-included so that their printed numbers carry 17 digits.
__graft_entry__.build() calls build_all() where /root/reference exists (the build container); the binaries travel to the
GPU box with the snapshot, where tests/test_reference_examples_gpu.py runs each pair and compares what they print.
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
BIN = os.path.join(ROOT, "tests", "cpp", "bin")
EXAMPLES = ["example-quadratic", "example-rosenbrock", "example-rosenbrock-box", "example-rosenbrock-bracketing",
            "example-rosenbrock-comparison"]
COMMON = ["g++", "-std=c++17", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-DSHIM_ACC=1",
          "-include", os.path.join(ROOT, "tests", "cpp", "precise_cout.h"), "-I", os.path.join(ROOT, "oracle", "eigen_shim"),
          "-I", os.path.join(ROOT, "oracle")]


def paths(name):
    return os.path.join(BIN, name + ".gpu"), os.path.join(BIN, name + ".ref")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in sources)


def build_all(force=False):
    """Returns the list of binaries built (empty where /root/reference does not exist: the prebuilt ones are kept)."""
    if not os.path.isdir(os.path.join(REF, "examples")):
        return []
    os.makedirs(BIN, exist_ok=True)
    lib = os.path.join(ROOT, "lbfgspp_amd")
    inc = os.path.join(ROOT, "include")
    deps = [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")] + \
           [os.path.join(inc, "LBFGSpp", f) for f in os.listdir(os.path.join(inc, "LBFGSpp"))] + \
           [os.path.join(ROOT, "oracle", "eigen_shim", "Eigen", "Core"), os.path.join(ROOT, "tests", "cpp", "precise_cout.h")]
    jobs = []
    for name in EXAMPLES:
        src = os.path.join(REF, "examples", name + ".cpp")
        gpu, ref = paths(name)
        if force or _stale(gpu, deps + [src]):
            # rpath relative to the binary: tests/cpp/bin -> lbfgspp_amd
            jobs.append(COMMON + ["-I", inc, src, "-o", gpu, "-L" + lib, "-llbfgsx", "-L/opt/rocm/lib", "-lamdhip64",
                                  "-Wl,-rpath,$ORIGIN/../../../lbfgspp_amd", "-Wl,-rpath,/opt/rocm/lib"])
        if force or _stale(ref, deps + [src]):
            jobs.append(COMMON + ["-I", os.path.join(REF, "include"), src, "-o", ref])
    # the solve-by-solve probe of the comparison loop (our own source, tests/cpp/cmp_probe.cpp), same two flavours
    psrc = os.path.join(ROOT, "tests", "cpp", "cmp_probe.cpp")
    pgpu, pref = paths("cmp_probe")
    if force or _stale(pgpu, deps + [psrc]):
        jobs.append(COMMON + ["-I", inc, psrc, "-o", pgpu, "-L" + lib, "-llbfgsx", "-L/opt/rocm/lib", "-lamdhip64",
                              "-Wl,-rpath,$ORIGIN/../../../lbfgspp_amd", "-Wl,-rpath,/opt/rocm/lib"])
    if force or _stale(pref, deps + [psrc]):
        jobs.append(COMMON + ["-I", os.path.join(REF, "include"), psrc, "-o", pref])
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(max(len(jobs), 1), os.cpu_count() or 1)) as ex:
        list(ex.map(subprocess.check_call, jobs))
    return [j[j.index("-o") + 1] for j in jobs]


if __name__ == "__main__":
    print("\n".join(build_all(force=True)))
