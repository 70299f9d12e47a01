"""Builds csrc/*.hip into libadcdiff.so (gfx950 only) next to this file.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the resulting
shared object travels to the GPU box with the source tree.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libadcdiff.so")
SOURCES = [("gemm.hip", []), ("gemm_tr.hip", []), ("elementwise.hip", ["-ffp-contract=off"]), ("unet.hip", []), ("train.hip", [])]
COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # the assembly main loop of the 4-wave GEMM is generated (register-allocated by hand in the generator)
    gen, inc = os.path.join(CSRC, "gen_w4_mainloop.py"), os.path.join(CSRC, "w4_mainloop.inc")
    if force or _stale(inc, [gen]):
        subprocess.run([sys.executable, gen, inc], check=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "unet_plan.h"), inc, os.path.join(os.path.dirname(HERE), "include", "adcdiff.h")]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_library(verbose=True))
