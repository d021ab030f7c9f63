"""Builds the HIP backend in-tree: csrc/evomg.hip -> csrc/libevomg.so for gfx950 (MI355X).

hipcc cross-compiles without a GPU. -ffp-contract=off keeps the arithmetic order of the kernels
explicit (bit-for-bit comparable with the CPU oracle); these stencils are bandwidth-bound, FMA
contraction would buy nothing.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libevomg.so")
SOURCES = ["evomg.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + ["../../include/evomg.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
