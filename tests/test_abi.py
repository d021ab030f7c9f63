"""CPU-side checks of the boundary: the HIP library loads without a GPU and exports every symbol
include/evomg.h declares; the ctypes mirror has the same struct sizes as the C header."""
import ctypes
import os
import re
import subprocess
import tempfile

import pytest

from evostencils_amd import abi, build

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "evomg.h")


@pytest.fixture(scope="module")
def libpath():
    return build.build()


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(evomg_[a-z_]+)\s*\(", text)))


def test_header_declares_what_the_wrapper_binds():
    assert declared_functions() == sorted(f"evomg_{s}" for s in abi.EXPORTED)


def test_library_loads_and_exports_every_declared_symbol(libpath):
    lib = ctypes.CDLL(libpath)
    for name in declared_functions():
        assert hasattr(lib, name), name
    lib.evomg_abi_version.restype = ctypes.c_int
    assert lib.evomg_abi_version() == abi.ABI_VERSION


def test_wrapper_binds_library(libpath):
    lib = abi.Library(libpath, "evomg", abi.EXPORTED)
    assert lib.fn["abi_version"]() == abi.ABI_VERSION


def test_struct_sizes_match_c_header():
    src = r'''
    #include <stdio.h>
    #include "evomg.h"
    int main(void) {
      printf("%zu %zu %zu %zu\n", sizeof(evomg_problem_t), sizeof(evomg_op_t), sizeof(evomg_result_t), sizeof(evomg_bench_t));
      return 0;
    }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), c, "-o", exe], check=True)
        sizes = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [ctypes.sizeof(abi.Problem), ctypes.sizeof(abi.Op), ctypes.sizeof(abi.Result), ctypes.sizeof(abi.Bench)]


def test_create_rejects_bad_descriptors_without_touching_the_gpu(libpath):
    lib = abi.Library(libpath, "evomg", abi.EXPORTED)
    p = abi.Problem()
    p.abi_version = 999
    with pytest.raises(abi.BackendError):
        abi.Handle(lib, p, 0)
    p.abi_version, p.dimension = abi.ABI_VERSION, 4
    with pytest.raises(abi.BackendError):
        abi.Handle(lib, p, 0)


def test_default_device_follows_the_launchers_local_rank(monkeypatch):
    """One process per GPU: torchrun's LOCAL_RANK, Open MPI's OMPI_COMM_WORLD_LOCAL_RANK (the reference is started with
    mpiexec), Slurm's SLURM_LOCALID ... decide the device, wrapped to the number of visible GPUs; without any of them the
    MPI rank does."""
    from evostencils_amd.code_generation.hip import ProgramGenerator

    class Lib:
        def __init__(self, n):
            self.fn = {"device_count": lambda: n}

    for name in ("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID"):
        monkeypatch.delenv(name, raising=False)
    assert ProgramGenerator._default_device(5, Lib(8)) == 5
    assert ProgramGenerator._default_device(11, Lib(8)) == 3          # second node of an 8-GPU-per-node job
    assert ProgramGenerator._default_device(11, Lib(0)) == 11         # no runtime: evomg_create reports it
    monkeypatch.setenv("OMPI_COMM_WORLD_LOCAL_RANK", "6")
    assert ProgramGenerator._default_device(14, Lib(8)) == 6
    monkeypatch.setenv("LOCAL_RANK", "2")                             # torchrun wins
    assert ProgramGenerator._default_device(14, Lib(8)) == 2
    monkeypatch.setenv("LOCAL_RANK", "-1")                            # not a rank: ignored, the next launcher's counts
    assert ProgramGenerator._default_device(14, Lib(8)) == 6
    monkeypatch.delenv("OMPI_COMM_WORLD_LOCAL_RANK")
    assert ProgramGenerator._default_device(14, Lib(8)) == 6          # ... or the MPI rank
    with pytest.raises(ValueError):
        ProgramGenerator._default_device(-1, Lib(8))


def test_constructing_a_generator_does_not_touch_the_hip_runtime(monkeypatch):
    """Lowering-only use (and everything before the first evaluation) must not initialise HIP: the device is resolved when the
    first handle is created, not in the constructor (a later fork would inherit a broken runtime, and
    GPU_MAX_HW_QUEUES has to be in the environment before the first HIP call)."""
    from evostencils_amd import problems
    from evostencils_amd.code_generation.hip import ProgramGenerator
    called = []

    class Lib:
        fn = {"device_count": lambda: called.append(1) or 8}

    gen = ProgramGenerator(problem=problems.poisson_2d(3, 5), library=Lib(), mpi_rank=3)
    assert gen._device_id is None and not called
    explicit = ProgramGenerator(problem=problems.poisson_2d(3, 5), library=Lib(), device_id=5)
    assert explicit._device_id == 5 and not called


def _run_python(code, **env):
    full = dict(os.environ, **env)
    return subprocess.run(["python", "-c", code], capture_output=True, text=True, env=full, cwd=REPO, timeout=120)


def test_a_host_without_rccl_is_reported_not_crashed(libpath):
    """include/evomg.h: evomg_rccl_version() is 0 and evomg_comm_create() EVOMG_ERR_UNSUPPORTED where librccl cannot be opened
    (the message was built from two dlerror() calls, the second of which returns NULL)."""
    code = ("import ctypes as C\n"
            f"lib = C.CDLL({libpath!r})\n"
            "lib.evomg_rccl_version.restype = C.c_int\n"
            "print('version', lib.evomg_rccl_version())\n"
            "comm = C.c_void_p()\n"
            "rc = lib.evomg_comm_create(0, 1, 0, b'/tmp/evomg_test_no_rccl.id', 100, C.byref(comm))\n"
            "lib.evomg_comm_last_error.restype = C.c_char_p\n"
            "lib.evomg_comm_last_error.argtypes = [C.c_void_p]\n"
            "print('rc', rc, lib.evomg_comm_last_error(None).decode())\n")
    out = _run_python(code, EVOMG_RCCL_ONLY="/nonexistent/librccl.so.1")
    assert out.returncode == 0, out.stderr
    assert "version 0" in out.stdout
    assert f"rc {abi.ERR_UNSUPPORTED} librccl not found" in out.stdout


def test_a_library_from_before_a_symbol_existed_asks_for_a_rebuild():
    """native.library() turns the AttributeError of a missing export into the readable 'rebuild' message."""
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "old.c")
        open(c, "w").write("int evomg_abi_version(void) { return 1; }\nint evomg_create(void) { return 1; }\n")
        so = os.path.join(d, "libold.so")
        subprocess.run(["gcc", "-shared", "-fPIC", c, "-o", so], check=True)
        out = _run_python("from evostencils_amd import native\n"
                          "try:\n    native.library()\nexcept RuntimeError as exc:\n    print('RuntimeError:', exc)\n", EVOMG_LIB=so)
    assert out.returncode == 0, out.stderr
    assert "RuntimeError:" in out.stdout and "rebuild with `python -m evostencils_amd.build`" in out.stdout


def test_register_budgets_the_documentation_states_hold_in_the_built_code_object():
    """DESIGN.md quotes occupancy figures; they are read off the gfx950 code object that ships (build.kernel_resources), not
    off a build somebody remembers (the cross kernel was documented at 161 VGPRs and shipped with 176)."""
    res = build.kernel_resources("impl_real")
    cross = {k: v for k, v in res.items() if "evomg_strip2d_cross" in k}
    assert len(cross) >= 16                           # (24 since round 6: the unit-weight forms)
    # the variants a Poisson cycle launches (power-of-two stencil, <.., RECIP = true>): at most 168 VGPRs = three waves per SIMD
    for name, r in cross.items():
        if name.split("EEv3Geo")[0].endswith("Lb1"):
            assert r["vgpr"] + r["agpr"] <= 168 and r["vgpr_spill"] == 0 and r["scratch"] == 0, (name, r)
    # the headline launches — the general power-of-two form and (round 6) the unit-weight form C2 runs: a bound, not an equality
    # (a compiler update must not fail the CPU suite); the figures as built are in DESIGN.md section 4: 166 / 164
    for tag in ("ILb1ELb1ELb1ELb1ELb0E", "ILb1ELb1ELb1ELb1ELb1E"):
        headline = next(v for k, v in cross.items() if tag in k)
        assert headline["vgpr"] <= 168 and headline["sgpr_spill"] == 0, (tag, headline)
    # the single-reduction CG kernels: nothing spills (a spill inside a latency chain is a round trip to memory per iteration)
    cg1 = {k: v for k, v in res.items() if "evomg_cg1_" in k}
    assert len(cg1) >= 14
    for name, r in cg1.items():
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0, (name, r)
    band129 = [v for k, v in cg1.items() if "band2dILi4ELi2ELi1ELi2E" in k]      # 129^2: sixteen workgroups of two waves (round 6)
    assert band129 and all(v["vgpr"] <= 256 for v in band129)     # at most one wave per SIMD: far inside the 512-register file
    slab = next(v for k, v in cg1.items() if "slab3d" in k)
    assert slab["vgpr"] <= 128                       # 1024 threads per workgroup
