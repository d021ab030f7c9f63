"""ctypes mirror of include/evomg.h and a thin object wrapper over a loaded C-ABI library.

The same wrapper drives the HIP backend (symbol prefix ``evomg_``, csrc/libevomg.so) and — from
tests/bench only — the CPU oracle (prefix ``oracle_``), because both consume the same op program.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

ABI_VERSION = 6
MAX_LEVELS = 24
MAX_DIM = 3
STENCIL_ENTRIES = 7
MAX_BLOCK_CELLS = 8
MAX_UNKNOWNS = 3

OK, ERR_INVALID, ERR_DEVICE, ERR_UNSUPPORTED = 0, 1, 2, 3
SOLVE_CONVERGED, SOLVE_MAXITER, SOLVE_DIVERGED, SOLVE_WALL_LIMIT = 0, 1, 2, 3
REAL_F64, COMPLEX_F64 = 0, 1
BC_DIRICHLET, BC_HELMHOLTZ_ROBIN = 0, 1
OUTER_MG_ITERATION, OUTER_PBICGSTAB = 0, 1
CGS_CG, CGS_BICGSTAB, CGS_FAS_SMOOTHER, CGS_CG_CLASSICAL = 0, 1, 2, 3
STENCIL_STAR, STENCIL_COMPACT = 0, 1
NONLINEAR_NONE, NONLINEAR_GAM_U_EXP_U, NONLINEAR_PROGRAM = 0, 1, 2
# reverse-Polish programs of a non-linear stencil entry (include/evomg.h: evomg_nl_program_t, ABI 6)
NL_MAX_OPS, NL_MAX_CONSTS, NL_MAX_STACK = 48, 8, 8
NL_END, NL_PUSH_U, NL_PUSH_CONST, NL_DUP = 0, 1, 2, 10
NL_ADD, NL_SUB, NL_MUL, NL_DIV, NL_NEG, NL_EXP, NL_SIN, NL_COS, NL_SQRT, NL_LOG, NL_TANH = 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26
NL_SINH, NL_COSH = 27, 28
# `with jacobi` slot handling (include/evomg.h EVOMG_JACOBI_*): TEXTBOOK = the sweep's result becomes the field;
# EXASTENCILS_V11 = the next slot is never advanced, as in the reference's pipeline (notebooks/tutorial.ipynb:3373)
JACOBI_TEXTBOOK, JACOBI_EXASTENCILS_V11 = 1, 2
JACOBI_SEMANTICS = {"textbook": JACOBI_TEXTBOOK, "exastencils-v1.1": JACOBI_EXASTENCILS_V11}
F_SOL, F_RHS, F_RES, F_CORR, F_CGS_RHS, F_COUNT = 0, 1, 2, 3, 4, 5
(OP_SMOOTH, OP_RESIDUAL, OP_RESTRICT, OP_PROLONG_ADD, OP_PROLONG_SET, OP_SET_ZERO, OP_COPY,
 OP_COARSE_SOLVE, OP_RICHARDSON) = range(1, 10)
# non-linear FAS vocabulary (include/evomg.h; exastencils_FAS.py:99-319)
(OP_FAS_SMOOTH, OP_FAS_RESIDUAL, OP_FAS_RESTRICT_SOLUTION, OP_FAS_COARSE_RHS, OP_FAS_CGS, OP_FAS_SUBTRACT) = range(16, 22)
PROF_LOOP_NORM = 64      # pseudo-opcode of profile_begin: the solver loop's own residual + norm pass (EVOMG_PROF_LOOP_NORM)
PART_SINGLE, PART_RED_BLACK = 0, 1

OP_NAMES = {OP_SMOOTH: "SMOOTH", OP_RESIDUAL: "RESIDUAL", OP_RESTRICT: "RESTRICT",
            OP_PROLONG_ADD: "PROLONG_ADD", OP_PROLONG_SET: "PROLONG_SET", OP_SET_ZERO: "SET_ZERO",
            OP_COPY: "COPY", OP_COARSE_SOLVE: "COARSE_SOLVE", OP_RICHARDSON: "RICHARDSON",
            OP_FAS_SMOOTH: "FAS_SMOOTH", OP_FAS_RESIDUAL: "FAS_RESIDUAL", OP_FAS_RESTRICT_SOLUTION: "FAS_RESTRICT_SOLUTION",
            OP_FAS_COARSE_RHS: "FAS_COARSE_RHS", OP_FAS_CGS: "FAS_CGS", OP_FAS_SUBTRACT: "FAS_SUBTRACT"}


class C64(C.Structure):
    _fields_ = [("re", C.c_double), ("im", C.c_double)]


class NlProgram(C.Structure):
    _fields_ = [("n_sigma", C.c_int32), ("n_jacobian", C.c_int32), ("sigma", C.c_uint8 * NL_MAX_OPS),
                ("jacobian", C.c_uint8 * NL_MAX_OPS), ("consts", C.c_double * NL_MAX_CONSTS)]


class Problem(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("dimension", C.c_int32), ("min_level", C.c_int32),
        ("max_level", C.c_int32), ("scalar_kind", C.c_int32), ("bc_kind", C.c_int32),
        ("outer_kind", C.c_int32), ("cgs_kind", C.c_int32), ("max_iterations", C.c_int32),
        ("cgs_max_iterations", C.c_int32), ("target_res_reduction", C.c_double),
        ("cgs_target_res_reduction", C.c_double),
        ("mg_stencil", (C64 * STENCIL_ENTRIES) * MAX_LEVELS),
        ("outer_stencil", C64 * STENCIL_ENTRIES),
        ("wavenumber", C.c_double),
        ("jacobi_semantics", C.c_int32), ("n_unknowns", C.c_int32),
        ("sys_stencil", (((C.c_double * 9) * MAX_UNKNOWNS) * MAX_UNKNOWNS) * MAX_LEVELS),
        ("nonlinear_kind", C.c_int32), ("stencil_kind", C.c_int32), ("nonlinear_gamma", C.c_double),
        ("fas_cgs_omega", C.c_double), ("nonlinear_program", NlProgram),
    ]


class Op(C.Structure):
    _fields_ = [
        ("opcode", C.c_int32), ("level", C.c_int32), ("dst", C.c_int32), ("src0", C.c_int32),
        ("src1", C.c_int32), ("partitioning", C.c_int32), ("block", C.c_int32 * MAX_DIM),
        ("unknowns", C.c_int32), ("block_other", C.c_int32 * MAX_DIM), ("block_third", C.c_int32), ("omega", C.c_double),
    ]

    def key(self):
        return (self.opcode, self.level, self.dst, self.src0, self.src1, self.partitioning,
                tuple(self.block), self.omega, self.unknowns, tuple(self.block_other), self.block_third)

    def __repr__(self):
        return (f"Op({OP_NAMES.get(self.opcode, self.opcode)}, L={self.level}, dst={self.dst}, "
                f"src0={self.src0}, src1={self.src1}, part={self.partitioning}, "
                f"block={tuple(self.block)}, omega={self.omega})")


class Result(C.Structure):
    _fields_ = [
        ("time_ms", C.c_double), ("conv_factor", C.c_double), ("n_iterations", C.c_double),
        ("init_res_norm", C.c_double), ("final_res_norm", C.c_double),
        ("status", C.c_int32), ("n_history", C.c_int32),
    ]


class Bench(C.Structure):
    _fields_ = [("ms", C.c_double), ("bytes", C.c_double), ("gbps", C.c_double),
                ("launches", C.c_int32), ("reserved", C.c_int32)]


def field_of(slot: int, unknown: int = 0) -> int:
    """Field id of `slot` (F_*) of the `unknown`-th unknown of a system of PDEs (EVOMG_FIELD in include/evomg.h)."""
    return slot + F_COUNT * unknown


def make_op(opcode, level, dst=0, src0=0, src1=0, partitioning=PART_SINGLE, block=(1, 1, 1), omega=1.0, unknowns=0,
            block_other=None, block_third=None) -> Op:
    op = Op()
    if block_third is not None:   # system smoother over three unknowns: the third listed unknown's block shape (packed, ABI 6)
        third = tuple(block_third) + (1,) * (MAX_DIM - len(block_third))
        op.block_third = int(third[0]) | (int(third[1]) << 8) | (int(third[2]) << 16)
    if block_other is not None:   # system smoother: the second unknown's block shape where it differs from `block`
        other = tuple(block_other) + (1,) * (MAX_DIM - len(block_other))
        for i in range(MAX_DIM):
            op.block_other[i] = int(other[i])
    op.unknowns = unknowns        # SMOOTH on a system: bit mask of the unknowns solved together (0 = scalar problem)
    op.opcode, op.level, op.dst, op.src0, op.src1 = opcode, level, dst, src0, src1
    op.partitioning = partitioning
    blk = tuple(block) + (1,) * (MAX_DIM - len(block))
    for i in range(MAX_DIM):
        op.block[i] = int(blk[i])
    op.omega = float(omega)
    return op


def ops_array(ops: Sequence[Op]):
    arr = (Op * max(len(ops), 1))()
    for i, op in enumerate(ops):
        arr[i] = op
    return arr


# every symbol include/evomg.h declares, without prefix
EXPORTED = ["create", "destroy", "last_error", "abi_version", "set_initial_field", "get_field",
            "set_field", "set_operators", "evaluate", "run_ops", "norm", "bench_op", "kernel_name",
            "profile_begin", "profile_end", "device_count", "time_iterations", "time_model", "time_model_known_coarse", "coarse_iterations",
            "comm_create", "comm_destroy", "comm_last_error", "allgather_fitness", "gather_fitness", "rccl_version"]
FITNESS_COLUMNS = 4      # EVOMG_FITNESS_COLUMNS: (candidate index, convergence factor, time ms, iterations)


class BackendError(RuntimeError):
    pass


class Library:
    """A loaded library exporting the C-ABI of include/evomg.h (`prefix`_`symbol`). The product instantiates it for
    csrc/libevomg.so only (native.py); test infrastructure binds other implementations of the same header with its own
    prefix and symbol subset (oracle/oracle.py)."""

    def __init__(self, path: str, prefix: str = "evomg", symbols: Sequence[str] = tuple(EXPORTED)):
        self.path, self.prefix = path, prefix
        self.lib = C.CDLL(path)
        self.fn = {}
        for s in symbols:
            self.fn[s] = getattr(self.lib, f"{prefix}_{s}")  # AttributeError if a symbol is missing
        dp = C.POINTER(C.c_double)
        f = self.fn
        f["create"].argtypes = [C.POINTER(Problem), C.c_int, C.POINTER(C.c_void_p)]
        f["create"].restype = C.c_int
        f["destroy"].argtypes = [C.c_void_p]
        f["destroy"].restype = None
        f["last_error"].argtypes = [C.c_void_p]
        f["last_error"].restype = C.c_char_p
        f["abi_version"].argtypes = []
        f["abi_version"].restype = C.c_int
        if "device_count" in f:
            f["device_count"].argtypes = []
            f["device_count"].restype = C.c_int
        for name in ("set_initial_field", "get_field", "set_field"):
            f[name].argtypes = [C.c_void_p, C.c_int, C.c_int, dp]
            f[name].restype = C.c_int
        f["set_operators"].argtypes = [C.c_void_p, C.POINTER(Problem)]
        f["set_operators"].restype = C.c_int
        f["evaluate"].argtypes = [C.c_void_p, C.POINTER(Op), C.c_int, C.c_int, C.c_double,
                                  C.POINTER(Result), dp, C.c_int]
        f["evaluate"].restype = C.c_int
        f["run_ops"].argtypes = [C.c_void_p, C.POINTER(Op), C.c_int]
        f["run_ops"].restype = C.c_int
        f["norm"].argtypes = [C.c_void_p, C.c_int, C.c_int, dp]
        f["norm"].restype = C.c_int
        if "bench_op" in f:
            f["bench_op"].argtypes = [C.c_void_p, C.POINTER(Op), C.c_int, C.c_int, C.POINTER(Bench)]
            f["bench_op"].restype = C.c_int
        if "profile_begin" in f:
            f["profile_begin"].argtypes = [C.c_void_p, C.c_int, C.c_int]
            f["profile_begin"].restype = C.c_int
            f["profile_end"].argtypes = [C.c_void_p, C.POINTER(Bench)]
            f["profile_end"].restype = C.c_int
        if "kernel_name" in f:
            f["kernel_name"].argtypes = [C.c_void_p, C.POINTER(Op), C.c_char_p, C.c_int]
            f["kernel_name"].restype = C.c_int
        if "time_iterations" in f:
            f["time_iterations"].argtypes = [C.c_void_p, C.POINTER(Op), C.c_int, C.c_int, dp, C.POINTER(C.c_int)]
            f["time_iterations"].restype = C.c_int
        if "time_model" in f:
            f["time_model"].argtypes = [C.c_void_p, C.POINTER(Op), C.c_int, C.c_int, C.c_int, dp, dp, dp]
            f["time_model"].restype = C.c_int
        if "time_model_known_coarse" in f:
            f["time_model_known_coarse"].argtypes = [C.c_void_p, C.POINTER(Op), C.c_int, C.c_int, C.c_int, C.c_double, dp, dp]
            f["time_model_known_coarse"].restype = C.c_int
            f["coarse_iterations"].argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
            f["coarse_iterations"].restype = C.c_int
        if "comm_create" in f:
            f["comm_create"].argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
            f["comm_create"].restype = C.c_int
            f["comm_destroy"].argtypes = [C.c_void_p]
            f["comm_destroy"].restype = None
            f["comm_last_error"].argtypes = [C.c_void_p]
            f["comm_last_error"].restype = C.c_char_p
            f["allgather_fitness"].argtypes = [C.c_void_p, dp, C.c_int, dp]
            f["allgather_fitness"].restype = C.c_int
            f["gather_fitness"].argtypes = [C.c_void_p, dp, C.c_int, dp, C.c_int]
            f["gather_fitness"].restype = C.c_int
            f["rccl_version"].argtypes = []
            f["rccl_version"].restype = C.c_int


@dataclass
class SolveOutcome:
    time_ms: float
    conv_factor: float
    n_iterations: float
    init_res_norm: float
    final_res_norm: float
    status: int
    history: Optional[np.ndarray] = None


class Handle:
    """One solver instance (all level fields allocated once) on one device."""

    def __init__(self, library: Library, problem: Problem, device_id: int = 0):
        self.L = library
        self.problem = problem
        self._h = C.c_void_p()
        rc = library.fn["create"](C.byref(problem), device_id, C.byref(self._h))
        if rc != OK:
            msg = library.fn["last_error"](None)
            raise BackendError(f"{library.prefix}_create failed with status {rc}: "
                               f"{msg.decode() if msg else ''}")
        self.dimension = problem.dimension
        self.is_complex = problem.scalar_kind == COMPLEX_F64

    def close(self):
        if self._h:
            self.L.fn["destroy"](self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != OK:
            msg = self.L.fn["last_error"](self._h)
            raise BackendError(f"{self.L.prefix}_{what} failed with status {rc}: {msg.decode() if msg else ''}")

    def last_error(self) -> str:
        msg = self.L.fn["last_error"](self._h)
        return msg.decode() if msg else ""

    def nodes(self, level: int) -> int:
        return (1 << level) + 1

    def field_shape(self, level: int):
        # numpy index order [i2][i1][i0]: x (i0) is the fastest dimension
        return (self.nodes(level),) * self.dimension

    def _host(self, level, array):
        dt = np.complex128 if self.is_complex else np.float64
        a = np.ascontiguousarray(array, dtype=dt)
        if a.shape != self.field_shape(level):
            raise ValueError(f"field shape {a.shape} != {self.field_shape(level)}")
        return a

    def set_initial_field(self, level, field, array):
        a = self._host(level, array)
        self._check(self.L.fn["set_initial_field"](self._h, level, field,
                                                   a.ctypes.data_as(C.POINTER(C.c_double))), "set_initial_field")

    def set_field(self, level, field, array):
        a = self._host(level, array)
        self._check(self.L.fn["set_field"](self._h, level, field,
                                           a.ctypes.data_as(C.POINTER(C.c_double))), "set_field")

    def get_field(self, level, field) -> np.ndarray:
        dt = np.complex128 if self.is_complex else np.float64
        out = np.empty(self.field_shape(level), dtype=dt)
        self._check(self.L.fn["get_field"](self._h, level, field,
                                           out.ctypes.data_as(C.POINTER(C.c_double))), "get_field")
        return out

    def set_operators(self, problem: Problem):
        self._check(self.L.fn["set_operators"](self._h, C.byref(problem)), "set_operators")
        self.problem = problem

    def run_ops(self, ops: Sequence[Op]):
        arr = ops_array(ops)
        self._check(self.L.fn["run_ops"](self._h, arr, len(ops)), "run_ops")

    def try_evaluate(self, ops: Sequence[Op], evaluation_samples=1, wall_limit_ms=0.0, history_capacity=0):
        """Returns (status_code, SolveOutcome|None) without raising for rejected programs."""
        arr = ops_array(ops)
        res = Result()
        hist = np.zeros(max(history_capacity, 1), dtype=np.float64)
        rc = self.L.fn["evaluate"](self._h, arr, len(ops), evaluation_samples, float(wall_limit_ms),
                                   C.byref(res), hist.ctypes.data_as(C.POINTER(C.c_double)) if history_capacity else None,
                                   history_capacity)
        if rc != OK:
            return rc, None
        h = hist[:res.n_history].copy() if history_capacity else None
        return rc, SolveOutcome(res.time_ms, res.conv_factor, res.n_iterations, res.init_res_norm,
                                res.final_res_norm, res.status, h)

    def evaluate(self, ops, evaluation_samples=1, wall_limit_ms=0.0, history_capacity=0) -> SolveOutcome:
        rc, out = self.try_evaluate(ops, evaluation_samples, wall_limit_ms, history_capacity)
        self._check(rc, "evaluate")
        return out

    def time_iterations(self, ops: Sequence[Op], n_iterations: int = 3):
        """(mean wall ms of one solver-loop iteration, iterations run) over exactly n_iterations iterations from the initial
        state, no convergence test (evomg_time_iterations): the contention-free time objective's probe."""
        arr = ops_array(ops)
        ms, done = C.c_double(), C.c_int()
        self._check(self.L.fn["time_iterations"](self._h, arr, len(ops), n_iterations, C.byref(ms), C.byref(done)), "time_iterations")
        return ms.value, done.value

    def time_model(self, ops: Sequence[Op], n_short: int = 2, n_long: int = 6, per_coarse: Optional[float] = None):
        """(one-off ms, ms per solver-loop iteration, ms per coarse-solver iteration) of the candidate on an idle device
        (evomg_time_model): time ~ fixed + per_iteration * its + per_coarse * K. `per_coarse` given: taken as known, the
        expensive third probe is skipped (evomg_time_model_known_coarse)."""
        arr = ops_array(ops)
        fixed, per_it, per_cg = C.c_double(), C.c_double(), C.c_double()
        if per_coarse is not None:
            self._check(self.L.fn["time_model_known_coarse"](self._h, arr, len(ops), n_short, n_long, float(per_coarse), C.byref(fixed),
                                                             C.byref(per_it)), "time_model_known_coarse")
            return fixed.value, per_it.value, float(per_coarse)
        self._check(self.L.fn["time_model"](self._h, arr, len(ops), n_short, n_long, C.byref(fixed), C.byref(per_it), C.byref(per_cg)),
                    "time_model")
        return fixed.value, per_it.value, per_cg.value

    def coarse_iterations(self) -> int:
        """Coarse-solver iterations of all coarse solves of the last evaluate (evomg_coarse_iterations)."""
        k = C.c_longlong()
        self._check(self.L.fn["coarse_iterations"](self._h, C.byref(k)), "coarse_iterations")
        return int(k.value)

    def norm(self, level, field) -> float:
        v = C.c_double()
        self._check(self.L.fn["norm"](self._h, level, field, C.byref(v)), "norm")
        return v.value

    def bench_op(self, op: Op, warmup=10, reps=100) -> Bench:
        b = Bench()
        self._check(self.L.fn["bench_op"](self._h, C.byref(op), warmup, reps, C.byref(b)), "bench_op")
        return b

    def profile_begin(self, opcode: int, level: int):
        self._check(self.L.fn["profile_begin"](self._h, opcode, level), "profile_begin")

    def profile_end(self) -> Bench:
        b = Bench()
        self._check(self.L.fn["profile_end"](self._h, C.byref(b)), "profile_end")
        return b

    def kernel_name(self, op: Op) -> str:
        buf = C.create_string_buffer(256)
        self._check(self.L.fn["kernel_name"](self._h, C.byref(op), buf, 256), "kernel_name")
        return buf.value.decode()


class Communicator:
    """The population's exchange step behind the C-ABI (include/evomg.h: evomg_comm_create / evomg_allgather_fitness /
    evomg_gather_fitness): one per process, bound to the process's GPU, RCCL underneath, no torch. `id_path`: a file every
    rank of the job can see and no other job uses (the RCCL unique id travels through it)."""

    def __init__(self, library: Library, rank: int, world: int, device_id: int, id_path: str, timeout_ms: int = 60000):
        self.L, self.rank, self.world = library, rank, world
        self._c = C.c_void_p()
        rc = library.fn["comm_create"](rank, world, device_id, id_path.encode(), timeout_ms, C.byref(self._c))
        if rc != OK:
            msg = library.fn["comm_last_error"](None)
            raise BackendError(f"evomg_comm_create failed with status {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if self._c:
            self.L.fn["comm_destroy"](self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _rows(self, rows):
        a = np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, FITNESS_COLUMNS)
        return a, a.ctypes.data_as(C.POINTER(C.c_double))

    def _check(self, rc, what):
        if rc != OK:
            msg = self.L.fn["comm_last_error"](self._c)
            raise BackendError(f"evomg_{what} failed with status {rc}: {msg.decode() if msg else ''}")

    def allgather(self, rows) -> np.ndarray:
        """rows: [n_local x 4] (the same n_local on every rank) -> [world * n_local x 4] in rank order, on every rank."""
        a, pa = self._rows(rows)
        out = np.empty((self.world * a.shape[0], FITNESS_COLUMNS), dtype=np.float64)
        self._check(self.L.fn["allgather_fitness"](self._c, pa, a.shape[0], out.ctypes.data_as(C.POINTER(C.c_double))), "allgather_fitness")
        return out

    def gather(self, rows, root: int = 0):
        """The same table on `root` only; None elsewhere."""
        a, pa = self._rows(rows)
        out = np.empty((self.world * a.shape[0], FITNESS_COLUMNS), dtype=np.float64) if self.rank == root else None
        po = out.ctypes.data_as(C.POINTER(C.c_double)) if out is not None else None
        self._check(self.L.fn["gather_fitness"](self._c, pa, a.shape[0], po, root), "gather_fitness")
        return out
