"""ctypes wrapper of the CPU oracle (oracle/libevomg_oracle.so). TEST INFRASTRUCTURE ONLY.

Reuses the product's ctypes struct definitions (evostencils_amd.abi) because the oracle consumes the
same op program; the product never imports this module.
"""
import os
import subprocess

from evostencils_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libevomg_oracle.so")


def build(force=False):
    """Compile oracle/evomg_oracle.c with gcc (recipe: oracle/Makefile)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return LIB_PATH


# the oracle implements the numerical subset of include/evomg.h (prefix oracle_)
ORACLE_EXPORTED = ["create", "destroy", "last_error", "abi_version", "set_initial_field", "get_field",
                   "set_field", "set_operators", "evaluate", "run_ops", "norm"]

_lib = None
_native_lib = None
NATIVE_LIB_PATH = os.path.join(_HERE, "libevomg_oracle_native.so")


def native_library() -> abi.Library:
    """The oracle compiled with -march=native ON THIS HOST (bench.py's cpu_baseline: SURVEY 8d asks for the CPU figure at
    -O3 -march=native). Same source, same -ffp-contract=off: same results. Raises if it cannot be built here."""
    global _native_lib
    if _native_lib is None:
        subprocess.run(["make", "-s", "-B", "-C", _HERE, "libevomg_oracle_native.so"], check=True)
        _native_lib = abi.Library(NATIVE_LIB_PATH, "oracle", ORACLE_EXPORTED)
    return _native_lib


def library() -> abi.Library:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = abi.Library(LIB_PATH, "oracle", ORACLE_EXPORTED)
    return _lib


def create(problem_description, parameters=None, jacobi_semantics=None, lib=None) -> abi.Handle:
    """Oracle solver instance with the problem's initial fields loaded. `jacobi_semantics`: None = the
    descriptor's, else "textbook" | "exastencils-v1.1" (abi.JACOBI_SEMANTICS). `lib`: another build of the oracle
    (native_library())."""
    h = abi.Handle(lib or library(), problem_description.to_c(parameters, jacobi_semantics), 0)
    problem_description.upload_initial_fields(h)
    return h
