#!/usr/bin/env python
"""Build oracle/_ref/libref_raster.so: the REFERENCE's rasterizer sources, compiled for the CPU.  TEST INFRASTRUCTURE.

Reads the three .cu files where they lie under /root/reference (never copied into the repository), rewrites ONLY the
CUDA triple-chevron launch syntax `kernel<<<grid, block>>>(args)` into `cuemu::launch(grid, block, kernel, args)`
(g++ cannot parse chevrons; every other token is left untouched), writes the rewritten translation units into
oracle/_ref/gen/ (git-ignored) and compiles them with g++ against the CUDA-on-CPU shim in oracle/cuda_cpu/ plus
oracle/ref_driver.cpp.  Flags: -O2 -ffp-contract=off, i.e. the reference's expressions evaluated as written with no
FMA contraction -- the same evaluation contract as oracle/raster_oracle.c and the HIP preprocess kernels.

/root/reference does not exist on the GPU box; the built .so travels there with the repository snapshot.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/gaussians/diff_gaussian_rasterization_depth_alpha"
SRC = os.path.join(REF, "cuda_rasterizer")
OUT = os.path.join(HERE, "_ref")
GEN = os.path.join(OUT, "gen")

LAUNCH = re.compile(r"(\b[A-Za-z_]\w*(?:<[^<>;]*>)?)\s*<<\s*<\s*([^;]*?)\s*>>\s*>\s*\(", re.S)


def rewrite(text: str) -> str:
    def sub(m):
        return f"cuemu::launch({m.group(2)}, {m.group(1)}, "
    out, n = LAUNCH.subn(sub, text)
    # CHECK_CUDA(, debug) appears after bare launches in rasterizer_impl.cu: an empty macro argument is legal
    return out, n


def main() -> int:
    if not os.path.isdir(SRC):
        print("reference checkout not present; keeping the prebuilt oracle/_ref (if any)")
        return 0
    os.makedirs(GEN, exist_ok=True)
    units = []
    total = 0
    for name in ("forward.cu", "backward.cu", "rasterizer_impl.cu"):
        with open(os.path.join(SRC, name)) as f:
            text, n = rewrite(f.read())
        total += n
        dst = os.path.join(GEN, name.replace(".cu", "_cpu.cpp"))
        with open(dst, "w") as f:
            f.write(f"// GENERATED from {os.path.join(SRC, name)} by oracle/ref_build.py -- do not commit\n" + text)
        units.append(dst)
    assert total == 8, f"expected 8 kernel launches in the reference, rewrote {total}"
    lib = os.path.join(OUT, "libref_raster.so")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-w",
           "-I", os.path.join(HERE, "cuda_cpu"), "-I", SRC, "-I", os.path.join(REF, "third_party", "glm"),
           "-o", lib] + units + [os.path.join(HERE, "cuda_cpu", "cuemu.cpp"), os.path.join(HERE, "ref_driver.cpp")]
    subprocess.check_call(cmd)
    print("built", lib)
    return 0


if __name__ == "__main__":
    sys.exit(main())
