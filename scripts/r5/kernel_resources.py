#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS table of every gfx950 kernel in the built objects (lbfgspp_amd/build/*.o).

    python scripts/r5/kernel_resources.py [--filter SUBSTR ...] [--min-vgpr N] [--scratch-only] [--tsv]

Reads the code objects the product links (no recompilation): .hip_fatbin section -> clang-offload-bundler --unbundle ->
llvm-readelf --notes (the AMDGPU metadata records).  waves/SIMD follows the gfx950 allocation rule of
MI355X_MICROARCH.md: 512 VGPRs per SIMD lane, granule 8, at most 8 waves.
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
KEYS = ("name", "vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size",
        "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size")


def code_object(obj, tmp):
    fb = os.path.join(tmp, os.path.basename(obj) + ".fb")
    co = os.path.join(tmp, os.path.basename(obj) + ".co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fb])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fb, "--output=" + co])
    return co


def kernels(co):
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    out, cur = [], {}
    for line in txt.splitlines():
        m = re.match(r"^\s+(?:- )?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "args" or k not in KEYS:
            continue
        if k in cur and k == "group_segment_fixed_size":
            out.append(cur)
            cur = {}
        cur[k] = v
    if cur:
        out.append(cur)
    return [k for k in out if "name" in k]


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), text=True, capture_output=True)
    return p.stdout.splitlines()


def short(name):
    # drop the argument list and the namespace: kx_rows<double, 10, 2, 3, true>
    s = re.sub(r"^void ", "", name)
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    s = s[:cut]
    return s.replace("lbfgsx::", "").replace("(anonymous namespace)::", "")


def waves_per_simd(vgpr, agpr):
    # .vgpr_count is the unified total (arch VGPRs + AGPRs, aligned); .agpr_count is the part of it that are AGPRs
    gran = (int(vgpr) + 7) // 8 * 8
    return min(8, 512 // max(gran, 8))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", nargs="*", default=[])
    ap.add_argument("--min-vgpr", type=int, default=0)
    ap.add_argument("--scratch-only", action="store_true")
    ap.add_argument("--tsv", action="store_true")
    ap.add_argument("--objs", nargs="*", default=sorted(glob.glob(os.path.join(ROOT, "lbfgspp_amd", "build", "*.o"))))
    a = ap.parse_args()
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for obj in a.objs:
            try:
                ks = kernels(code_object(obj, tmp))
            except subprocess.CalledProcessError:
                continue
            dn = demangle([k["name"] for k in ks])
            for k, d in zip(ks, dn):
                rows.append((os.path.basename(obj), short(d), int(k.get("vgpr_count", 0)), int(k.get("agpr_count", 0) or 0),
                             int(k.get("sgpr_count", 0)), int(k.get("private_segment_fixed_size", 0)),
                             int(k.get("group_segment_fixed_size", 0)), int(k.get("vgpr_spill_count", 0) or 0)))
    rows = [r for r in rows if r[2] >= a.min_vgpr and (not a.scratch_only or r[5] > 0)
            and (not a.filter or any(f in r[1] for f in a.filter))]
    rows.sort(key=lambda r: (r[0], r[1]))
    sep = "\t" if a.tsv else "  "
    print(sep.join(["object", "kernel", "vgpr", "agpr", "sgpr", "scratch_B", "lds_B", "vgpr_spills", "waves/SIMD"]))
    for r in rows:
        print(sep.join([r[0], r[1]] + [str(x) for x in r[2:]] + [str(waves_per_simd(r[2], r[3]))]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
