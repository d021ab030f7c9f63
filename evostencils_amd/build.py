"""Builds the HIP backend in-tree: csrc/*.hip -> csrc/libevomg.so for gfx950 (MI355X).

hipcc cross-compiles without a GPU. -ffp-contract=off keeps the arithmetic order of the kernels
explicit (bit-for-bit comparable with the CPU oracle); these stencils are bandwidth-bound, FMA
contraction would buy nothing.

One object per translation unit (the C-ABI, and one per implementation class: real, complex, systems, FAS),
compiled in parallel, then one link: a change to one implementation recompiles one object.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libevomg.so")
OBJ_DIR = os.path.join(CSRC, "obj")
SOURCES = ["evomg.hip", "impl_real.hip", "impl_complex.hip", "impl_system.hip", "impl_fas.hip", "evomg_collective.hip"]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fno-gpu-rdc", "-Wall",
          "-Wno-unused-function", "-fvisibility=hidden"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-ldl"]


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _headers():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".inl"))] + \
           [os.path.join(HERE, "..", "include", "evomg.h"), os.path.abspath(__file__)]


def _newer(deps, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")


def _stale():
    return _newer([os.path.join(CSRC, s) for s in _sources()] + _headers(), LIB)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = _headers()
    todo = [s for s in _sources() if force or _newer([os.path.join(CSRC, s)] + headers, _obj(s))]

    def compile_one(src):
        cmd = [hipcc] + CFLAGS + ["-c", os.path.join(CSRC, src), "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1) or 1) as pool:
        list(pool.map(compile_one, todo))
    cmd = [hipcc] + LDFLAGS + ["-o", LIB] + [_obj(s) for s in _sources()]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)


def kernel_resources(unit="impl_real"):
    """Register and LDS use of every kernel of one translation unit AS BUILT: {mangled name: {"vgpr", "agpr", "sgpr",
    "vgpr_spill", "sgpr_spill", "lds", "scratch"}} from the metadata notes of the gfx950 code object inside csrc/obj/<unit>.o
    (llvm-objcopy dumps the fat binary, clang-offload-bundler takes the device part out, llvm-readelf prints the notes).
    What the documentation says about occupancy is checked against this (tests/test_abi.py)."""
    import re
    import tempfile
    llvm = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")
    obj = _obj(unit + ".hip")
    if not os.path.exists(obj):
        build()
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "device.co")
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", obj], check=True)
        subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    keys = {".vgpr_count": "vgpr", ".agpr_count": "agpr", ".sgpr_count": "sgpr", ".vgpr_spill_count": "vgpr_spill",
            ".sgpr_spill_count": "sgpr_spill", ".group_segment_fixed_size": "lds", ".private_segment_fixed_size": "scratch"}
    out, current = {}, {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*(\.[a-z_]+):\s*(\S+)", line)
        if not m:
            continue
        key, value = m.group(1), m.group(2)
        if key == ".name" and "name" not in current:           # (arguments carry a .name too: they come before the kernel's own keys end)
            current["name"] = value
        elif key in keys:
            current[keys[key]] = int(value)
        elif key == ".wavefront_size":                          # last key of a kernel's map
            if "name" in current:
                out[current["name"]] = {k: v for k, v in current.items() if k != "name"}
            current = {}
    return out
