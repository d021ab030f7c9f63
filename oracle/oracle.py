"""ctypes wrapper of the CPU oracle (oracle/libevomg_oracle.so). TEST INFRASTRUCTURE ONLY.

Reuses the product's ctypes struct definitions (evostencils_amd.abi) because the oracle consumes the
same op program; the product never imports this module.
"""
import os
import subprocess

from evostencils_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# EVOMG_ORACLE_LIB: another build of the oracle (`make -C oracle sanitize` -> libevomg_oracle_asan.so, run under LD_PRELOAD=libasan)
LIB_PATH = os.path.abspath(os.environ["EVOMG_ORACLE_LIB"]) if os.environ.get("EVOMG_ORACLE_LIB") else os.path.join(_HERE, "libevomg_oracle.so")


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


_tuned_lib = None
TUNED_LIB_PATH = os.path.join(_HERE, "libevomg_cpu_tuned.so")


def tuned_poisson2d(lo, hi, u0, rhs, pre=1, post=1, omega=1.0, tol=1e-12, maxit=100, cgs_maxit=1000, cgs_tol=1e-12):
    """The TUNED CPU solver (oracle/evomg_cpu_tuned.c: plain OpenMP, not bit-comparable) on the 2D Poisson problem with
    initial guess / boundary values `u0` and right-hand side `rhs` on level `hi`: V(pre, post) red-black cycle, CG on level
    `lo`. Built on this host on first use. Returns (solution, residual history, coarse-solver iterations, seconds)."""
    import ctypes as C
    import time
    import numpy as np
    global _tuned_lib
    if _tuned_lib is None:
        subprocess.run(["make", "-s", "-B", "-C", _HERE, "libevomg_cpu_tuned.so"], check=True)
        _tuned_lib = C.CDLL(TUNED_LIB_PATH)
        dp = C.POINTER(C.c_double)
        _tuned_lib.evomg_cpu_tuned_poisson2d.argtypes = [C.c_int, C.c_int, dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                                          C.c_int, C.c_double, dp, C.c_int, C.POINTER(C.c_longlong)]
        _tuned_lib.evomg_cpu_tuned_poisson2d.restype = C.c_int
    dp = C.POINTER(C.c_double)
    u = np.ascontiguousarray(u0, dtype=np.float64).copy()
    f = np.ascontiguousarray(rhs, dtype=np.float64)
    hist = np.zeros(maxit + 1)
    k = C.c_longlong()
    t0 = time.time()
    its = _tuned_lib.evomg_cpu_tuned_poisson2d(lo, hi, u.ctypes.data_as(dp), f.ctypes.data_as(dp), pre, post, omega, tol, maxit,
                                               cgs_maxit, cgs_tol, hist.ctypes.data_as(dp), len(hist), C.byref(k))
    seconds = time.time() - t0
    if its < 0:
        raise RuntimeError("evomg_cpu_tuned_poisson2d: bad arguments")
    return u, hist[:its + 1], int(k.value), seconds
