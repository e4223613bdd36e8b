"""Build recipe for the native libraries (in-tree, gfx950 only).

    liblbfgsx.so         hipcc: HIP kernels + low-level C ABI (include/lbfgsx.h)
    liblbfgsx_solver.so  g++  : drop-in C++ solver templates instantiated for the built-in objectives
                                (include/lbfgsx_solver.h); links liblbfgsx.so
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# -ffp-contract=off: element-wise statements must round exactly like the reference's separate
# multiply/add expressions; FMAs are used only where written explicitly (error-free transformations).
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
             "-Wno-unused-result"]
# Host templates (line searches, BKLDLT, the sequential GCP form): x86-64-v3 so that std::fma -- the error-free product of
# the double-double sums -- is the hardware instruction instead of a libm call; -ffp-contract=off keeps every other
# expression un-fused (the parity contract of DESIGN.md section 2: no contraction anywhere).
CXX_FLAGS = ["-std=c++17", "-O3", "-march=x86-64-v3", "-ffp-contract=off", "-fPIC", "-shared", "-Wall"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _glob(d, exts):
    out = []
    for root, _, files in os.walk(d):
        out += [os.path.join(root, f) for f in files if f.endswith(exts)]
    return out


def _compile_one(args):
    src, obj, flags, verbose = args
    cmd = [HIPCC] + flags + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return obj


def build(force=False, verbose=False):
    inc = os.path.join(os.path.dirname(HERE), "include")
    hip_src = sorted(f for f in _glob(CSRC, (".hip",)))
    common = _glob(CSRC, (".cuh", ".hpp", ".map")) + _glob(inc, (".h",))
    lib = os.path.join(HERE, "liblbfgsx.so")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in HIP_FLAGS if f != "-shared"]
    # one object per translation unit, stale ones compiled concurrently (hipcc handles a list of sources serially)
    jobs = []
    objs = []
    for src in hip_src:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + common):
            jobs.append((src, obj, cflags, verbose))
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(_compile_one, jobs))
    deps = objs + _glob(CSRC, (".cpp",)) + common
    if force or jobs or _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + \
              ["-o", lib, "-Wl,--version-script=" + os.path.join(CSRC, "export.map")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    sol = os.path.join(HERE, "liblbfgsx_solver.so")
    if force or _stale(sol, deps + [lib]):
        cmd = ["g++"] + CXX_FLAGS + [os.path.join(CSRC, "solver_capi.cpp"), "-o", sol, "-L" + HERE, "-llbfgsx",
                                     "-Wl,-rpath,$ORIGIN", "-Wl,--version-script=" + os.path.join(CSRC, "export.map")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return lib, sol


if __name__ == "__main__":
    print(build(force=True, verbose=True))
