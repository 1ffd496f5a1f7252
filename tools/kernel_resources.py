#!/usr/bin/env python3
"""Registers, scratch and static LDS of every kernel of the library, from the gfx950 code objects hipcc emits (no device needed):
compiles each csrc/*.hip with -save-temps and reads the kernel descriptors' metadata. usage: tools/kernel_resources.py > profiles/<tag>_kernel_resources.txt"""
import glob
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "openvslam_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics"]   # the Makefile's


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    out = [re.sub(r"\(.*", "", o).replace("void ", "").strip() for o in out]
    return [o or n for o, n in zip(out, names)]   # (extern "C" kernels have nothing to demangle)


def collect(files=None):
    """[(file, kernel, vgpr, agpr, sgpr, scratch bytes, static LDS bytes, max workgroup size)] for csrc/*.hip (or the named files)."""
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
            if files is not None and os.path.basename(src) not in files:
                continue
            base = os.path.splitext(os.path.basename(src))[0]
            extra = {"pose_opt": ["-mllvm", "-disable-machine-licm"], "match_hamming": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}.get(base, [])   # the Makefile's per-file flags
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-c", src, "-o", os.path.join(td, base + ".o"), "-save-temps=obj"], cwd=CSRC, check=True,
                           stderr=subprocess.DEVNULL)
            asm = os.path.join(td, base + "-hip-amdgcn-amd-amdhsa-gfx950.s")
            if not os.path.exists(asm):
                continue
            txt = open(asm).read()
            for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
                f = {k: v for k, v in re.findall(r"\.(\w+):\s+(\S+)", blk)}
                rows.append((base, f.get("name", "?"), int(f.get("vgpr_count", 0)), int(f.get("agpr_count", 0)), int(f.get("sgpr_count", 0)),
                             int(f.get("private_segment_fixed_size", 0)), int(f.get("group_segment_fixed_size", 0)),
                             int(f.get("max_flat_workgroup_size", 0))))
    names = demangle([r[1] for r in rows])
    return [(r[0], n) + r[2:] for r, n in zip(rows, names)]


def main():
    rows = collect()
    print("# hipcc %s; vgpr = unified VGPR + AGPR count; waves/SIMD = floor(512 / vgpr) capped at 8 (LDS may bind lower)" % " ".join(FLAGS))
    print("%-18s %-58s %5s %5s %5s %8s %9s %7s %10s" % ("file", "kernel", "vgpr", "agpr", "sgpr", "scratch", "lds_stat", "max_wg", "waves/SIMD"))
    for (base, n, v, a, s, p, g, w) in rows:
        tot = max(v, 1)
        print("%-18s %-58s %5d %5d %5d %8d %9d %7d %10d" % (base, n[:58], v, a, s, p, g, w, min(8, 512 // tot)))


if __name__ == "__main__":
    main()
