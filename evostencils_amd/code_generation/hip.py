"""``ProgramGenerator`` — drop-in replacement for the reference's backend class
``evostencils.code_generation.exastencils.ProgramGenerator`` (exastencils.py:39-1147) behind the
duck-typed interface the optimizer uses (callers: optimization/program.py:115-141,
407,439,805,853,891-895,909-926; scripts/optimize.py:55-68).

Where the reference writes ExaSlang, starts the ExaStencils JVM twice, runs `make` and executes the
generated binary (exastencils.py:485-537), this class lowers the candidate to an op program and
calls ``evomg_evaluate`` (include/evomg.h) on one MI355X. There is no CPU fallback: without the HIP
library / a GPU the constructor raises.
"""
import math
import os
from typing import List, Optional

from evostencils_amd import abi, problems
from evostencils_amd.code_generation import exaslang, lowering, opprogram
from evostencils_amd.ir import base, system


class Field:
    """exastencils.py:29-36"""

    def __init__(self, name=None, level=None, cycle_storage=None):
        self.name, self.level, self.cycle_storage = name, level, cycle_storage

    def to_exa(self):
        return f"{self.name}@{self.level}"


class CycleStorage:
    """Names of the arrays of one level (exastencils.py:14-26). Opaque to the optimizer."""

    def __init__(self, equations, fields, grids):
        self.grid = grids
        self.solution = [Field(f"{s.name}", g.level, self) for g, s in zip(grids, fields)]
        self.rhs = []
        for f, g in zip(fields, grids):
            for eq in equations:
                if eq.level == g.level and eq.associated_field == f:
                    self.rhs.append(Field(f"{eq.rhs_name}", g.level, self))
                    break
        self.residual = [Field(f"gen_residual_{s.name}", g.level, self) for g, s in zip(grids, fields)]
        self.correction = [Field(f"gen_error_{s.name}", g.level, self) for g, s in zip(grids, fields)]


class ProgramGenerator:
    """Same constructor arguments as the reference (exastencils.py:40-43); the ExaStencils-specific
    ones (compiler path, platform file, compiler timeouts) are accepted and ignored. Either pass the
    reference-style `base_path` + `settings_path` + `knowledge_path` of one of the known example
    problems, or pass `problem=` (a problems.ProblemDescription) directly.

    Extra keyword-only arguments: `device_id` (default: LOCAL_RANK or mpi_rank), `library` (an
    abi.Library; default: the in-tree HIP library, loaded on first use), `jacobi_semantics`:

      "exastencils-v1.1" (default)  what the reference's pipeline does: a `with jacobi` sweep (every smoother with
          Single partitioning) writes the next slot of its field and that slot is never advanced, so the sweep costs
          its time but leaves the field unchanged. The advance patch (exastencils.py:338-359) searches the debug
          layer 4 for "[next]"; with the ExaStencils version the reference pins (README.md:19-21) it evidently never
          fires: the one numeric the reference records (notebooks/tutorial.ipynb:3373) is reproduced to the last
          printed digit this way and under no other convention (oracle/APPB_MATRIX.md).
      "textbook"  the sweep's result becomes the field (the IR / LFA meaning, and what that patch intends).

    `coarse_cg`: how `solver_cgs = "CG"` runs — "single-reduction" (default: include/evomg.h EVOMG_CGS_CG, both inner
    products of an iteration in one reduction where such a kernel exists) or "classical" (EVOMG_CGS_CG_CLASSICAL, the textbook
    recurrence everywhere). The same Krylov process either way; no number the reference records depends on it."""
    COARSE_COST_SAMPLES = 3      # full three-probe time models a generator runs before it takes the coarse-iteration cost as known
    COARSE_COST_REFRESH = 32     # ... and after how many two-probe models it measures that cost again (median of the last five)


    def __init__(self, absolute_compiler_path: Optional[str] = None, base_path: Optional[str] = None,
                 settings_path: Optional[str] = None, knowledge_path: Optional[str] = None,
                 platform_path: Optional[str] = None, mpi_rank=0, solution_equations=None,
                 cycle_name="gen_mgCycle", model_based_estimation=False, evaluation_timeout=300,
                 code_generation_timeout=300, c_compiler_timeout=120, solver_iteration_limit=None, *,
                 problem: Optional[problems.ProblemDescription] = None, device_id: Optional[int] = None,
                 library: Optional[abi.Library] = None, jacobi_semantics: str = "exastencils-v1.1",
                 coarse_cg: str = "single-reduction"):
        if isinstance(solution_equations, str):
            solution_equations = [solution_equations]
        self._average_generation_time = 0
        self._counter = 0
        self.timeout_evaluate = evaluation_timeout
        self.timeout_exastencils_compiler = code_generation_timeout
        self.timeout_c_compiler = c_compiler_timeout
        self._solver_iteration_limit = solver_iteration_limit
        self._absolute_compiler_path = absolute_compiler_path
        self._base_path = base_path
        self._platform_path = platform_path
        self._settings_path = settings_path
        self._knowledge_path = knowledge_path
        self._mpi_rank = mpi_rank
        self._cycle_name = cycle_name
        # exastencils.py:64-70: no `with jacobi` prefix under model-based estimation
        self._use_jacobi_prefix = not model_based_estimation
        self._solution_equations = solution_equations
        if problem is None:
            if base_path is None or settings_path is None or knowledge_path is None:
                raise ValueError("either `problem` or base_path/settings_path/knowledge_path is required")
            # the problem files themselves are the source (problem_files.py; the reference needs an ExaStencils run
            # and its debug layer-3 output for this, parser.py:25-143); problems that enter below layer 2
            # (Helmholtz: complex fields, Robin boundaries, Krylov outer solver) come from the built-in descriptors
            from evostencils_amd import problem_files
            knowledge = problem_files.read_key_values(os.path.join(base_path, knowledge_path))
            settings = problem_files.resolve_settings(problem_files.read_key_values(os.path.join(base_path, settings_path)))
            config = settings.get("configName")
            l2_path = os.path.join(base_path, settings.get("basePathPrefix", "."), settings.get("l2file", f"{config}.exa2"))
            problem = None
            try:
                # layer-2 files, or the layer-3 / layer-4 pair of a Helmholtz-type problem
                problem = problem_files.load_problem(base_path, settings_path, knowledge_path)
            except problem_files.UnsupportedProblem:
                # outside the front end's subset: the built-in descriptor of the same configuration, if there is one
                if config not in problems.BY_NAME:
                    raise
            if problem is not None:
                pass
            elif config in problems.BY_NAME:
                problem = problems.BY_NAME[config](int(knowledge["minLevel"]), int(knowledge["maxLevel"]))
            else:
                raise RuntimeError(f"problem {config!r} has no layer-2 file and is not a built-in descriptor "
                                   f"(known: {sorted(problems.BY_NAME)})")
            if int(knowledge["dimensionality"]) != problem.dimension:
                raise RuntimeError("dimensionality mismatch between knowledge file and problem")
        if jacobi_semantics not in abi.JACOBI_SEMANTICS:
            raise ValueError(f"jacobi_semantics must be one of {sorted(abi.JACOBI_SEMANTICS)}")
        if coarse_cg not in ("single-reduction", "classical"):
            raise ValueError("coarse_cg must be 'single-reduction' or 'classical'")
        import dataclasses
        problem = dataclasses.replace(problem, jacobi_semantics=abi.JACOBI_SEMANTICS[jacobi_semantics],
                                      parameters=dict(problem.parameters))
        if problem.cgs_kind in (abi.CGS_CG, abi.CGS_CG_CLASSICAL):
            problem = dataclasses.replace(problem, cgs_kind=abi.CGS_CG if coarse_cg == "single-reduction" else abi.CGS_CG_CLASSICAL)
        self._jacobi_semantics = jacobi_semantics
        self._problem = problem
        self._original_min_level = problem.min_level
        self._original_max_level = problem.max_level
        # resolved on first use (_ensure_handle): asking the library for the device count initialises the HIP runtime, which a
        # generator that only lowers must not do (a later fork would inherit a broken runtime), and GPU_MAX_HW_QUEUES
        # (evostencils_amd/__init__.py) only takes effect if it is set before that first call
        self._device_id = device_id
        self._library = library
        self._handle = None
        self._handle_levels = None
        self._handle_params = None
        self._solver_cache = {}
        self._verified = {}
        self._cycle_functions = {}       # text of a generated cycle function -> (level, window min, op program)
        self.last_coarse_iterations = None      # coarse-solver iterations of the last evaluation (time model input)
        self._coarse_iteration_ms = {}           # (min_level, max_level) -> {samples, the settled cost of one coarse-solver iteration, uses since}
        self._set_problem(problem)

    # ---- read-only attributes the optimizer reads (scripts/optimize.py:60-68) ------------------
    uses_FAS = property(lambda self: False)
    absolute_compiler_path = property(lambda self: self._absolute_compiler_path)
    knowledge_path = property(lambda self: self._knowledge_path)
    settings_path = property(lambda self: self._settings_path)
    platform_path = property(lambda self: self._platform_path)
    base_path = property(lambda self: self._base_path)
    problem_name = property(lambda self: self._problem.name)
    compiler_available = property(lambda self: True)
    dimension = property(lambda self: self._problem.dimension)
    finest_grid = property(lambda self: self._finest_grid)
    equations = property(lambda self: self._equations)
    operators = property(lambda self: self._operators)
    fields = property(lambda self: self._fields)
    coarsening_factor = property(lambda self: self._coarsening_factor)
    min_level = property(lambda self: self._problem.min_level)
    max_level = property(lambda self: self._problem.max_level)
    solution_equations = property(lambda self: self._solution_equations)
    mpi_rank = property(lambda self: self._mpi_rank)
    solver_iteration_limit = property(lambda self: self._solver_iteration_limit)
    problem = property(lambda self: self._problem)
    jacobi_semantics = property(lambda self: self._jacobi_semantics)

    def _set_problem(self, problem):
        self._problem = problem
        self._equations, self._operators, self._fields = problem.l2_information()
        d, L = problem.dimension, problem.max_level
        self._coarsening_factor = [(2,) * d for _ in self._fields]
        step = tuple((hi_ - lo_) / (2 ** L) for lo_, hi_ in zip(problem.domain_lower[:d], problem.domain_upper[:d]))
        self._finest_grid = [base.Grid((2 ** L,) * d, step, L) for _ in self._fields]

    # ---- storage handles (exastencils.py:586-592) -------------------------------------------------
    def generate_storage(self, min_level: int, max_level: int, finest_grids: List[base.Grid]):
        storage, grid = [], finest_grids
        for _ in range(min_level, max_level + 1):
            storage.append(CycleStorage(self.equations, self.fields, grid))
            grid = system.get_coarse_grid(grid, self.coarsening_factor)
        return storage

    # ---- native handle ------------------------------------------------------------------------------
    def _lib(self) -> abi.Library:
        if self._library is None:
            from evostencils_amd import native
            self._library = native.library()     # raises if the HIP library is missing
        return self._library

    @staticmethod
    def _default_device(mpi_rank, library=None):
        """One process per GPU: the rank's index ON ITS NODE, as the launcher exports it (torchrun: LOCAL_RANK; Open MPI,
        MVAPICH2, Intel MPI / MPICH hydra, Slurm — the reference is started with `mpiexec -n N`, README.md), else the MPI
        rank; modulo the number of GPUs the process sees, so that ranks beyond one node's GPUs wrap around instead of
        asking for a device that does not exist."""
        local = mpi_rank
        for name in ("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID"):
            if os.environ.get(name, "").isdigit():          # a negative or malformed value is not a rank
                local = int(os.environ[name])
                break
        if not isinstance(local, int) or local < 0:
            raise ValueError(f"local rank {local!r} is not a device index")
        count = 0
        if library is None:
            try:
                from evostencils_amd import native
                library = native.library()
            except Exception:      # noqa: BLE001 — not built: the constructor must still work for lowering-only use
                library = None
        fn = getattr(library, "fn", {}) if library is not None else {}
        if "device_count" in fn:
            try:
                count = int(fn["device_count"]())
            except Exception:      # noqa: BLE001 — no usable runtime: keep the plain index, evomg_create reports the failure
                count = 0
        return local % count if count > 0 else local

    def _ensure_handle(self, min_level, max_level, parameters=None):
        """One native solver instance per (level window, problem); operators are re-targeted in place
        when only PDE parameters (k) change."""
        key = (self._problem.name, min_level, max_level)
        params = dict(self._problem.parameters)
        if parameters:
            params.update(parameters)
        frozen = tuple(sorted(params.items()))
        if self._device_id is None:
            self._device_id = self._default_device(self._mpi_rank, self._lib())
        if self._handle is None or self._handle_levels != key:
            if self._handle is not None:
                self._handle.close()
            window = self._problem.with_levels(min_level, max_level)
            self._handle = abi.Handle(self._lib(), window.to_c(params), self._device_id)
            window.upload_initial_fields(self._handle, max_level)
            self._handle_levels, self._handle_params, self._window = key, frozen, window
        elif self._handle_params != frozen:
            self._handle.set_operators(self._window.to_c(params))
            self._handle_params = frozen
        return self._handle

    def initialize_code_generation(self, min_level: int, max_level: int):
        """exastencils.py:445-466: one-time setup for a level range. Here: allocate the device fields."""
        self._ensure_handle(min_level, max_level)
        self._counter += 1
        return None

    def reinitialize(self, min_level, max_level, global_expressions):
        """exastencils.py:196-215: problem-size bump ("generalisation"); `global_expressions` replaces
        PDE parameters (e.g. {'k': 160.0})."""
        new = self._problem.with_levels(min_level, max_level)
        for name, value in (global_expressions or {}).items():
            new.parameters[name] = value
        self._set_problem(new)

    # ---- lowering -----------------------------------------------------------------------------------
    def _names(self):
        return exaslang.FieldNames([f.name for f in self.fields],
                                   [next(eq.rhs_name for eq in self.equations if eq.associated_field == f)
                                    for f in self.fields])

    def lower(self, expression, min_level, max_level, use_global_weights=False):
        return lowering.lower(expression, self.fields, min_level, max_level, self._cycle_name, use_global_weights)

    def generate_multigrid(self, expression, storages, min_level: int, max_level: int, use_global_weights=False):
        """Statement text only (exastencils.py:684-925). `storages` is accepted for signature parity."""
        stmts = self.lower(expression, min_level, max_level, use_global_weights)
        return exaslang.render(stmts, self._names(), max_level, self.dimension, self._use_jacobi_prefix)

    def generate_cycle_function(self, expression, storages, min_level: int, level, max_level: int,
                                use_global_weights=False):
        """exastencils.py:318-336: `Function <cycle_name>@<level> { ... }`."""
        program = f"Function {self._cycle_name}@{level} {{\n"
        stmts = self.lower(expression, min_level, max_level, use_global_weights)
        program += exaslang.render(stmts, self._names(), max_level, self.dimension, self._use_jacobi_prefix)
        program += "}\n\n"
        if not use_global_weights:
            # multi-run optimisation hands this text back as part of `solver_program` (optimization/program.py:891-898):
            # remember what it means, so that it never has to be parsed
            try:
                self._verify_operators(stmts)
                self._cycle_functions[program] = (level, min_level, opprogram.encode(stmts))
            except lowering.UnsupportedExpression:
                pass
        return program

    def generate_op_program(self, expression, min_level, max_level, solver_program: str = "") -> List[abi.Op]:
        """Op program of ONE multigrid cycle on the problem's finest level, as the generated solver would run it.

        Whole-hierarchy optimisation (`min_level`..`max_level` = the problem's levels, empty `solver_program`): the
        lowering of `expression`. Multi-run optimisation (optimization/program.py:811-898): `expression` covers the
        window `min_level`..`max_level` only; the levels above come from the cycle functions in `solver_program`
        (texts this generator produced earlier), each of which calls `gen_mgCycle@<its coarsest level>()`; below the
        window the template's own functions remain, i.e. the problem's generated reference cycle
        (exastencils.py:927-969 strips only the levels min_level+1..max). Calls are expanded in place."""
        top = self._problem.max_level
        stmts = self.lower(expression, min_level, top if (solver_program or max_level < top) else max_level)
        self._verify_operators(stmts)
        functions = self._functions_of(solver_program)
        functions[max_level] = (min_level, opprogram.encode(stmts))
        if not solver_program and max_level == top and min_level == self._problem.min_level:
            return functions[max_level][1]
        if top not in functions:
            raise lowering.UnsupportedExpression(f"no cycle function for level {top}: solver_program must hold the "
                                                 f"functions of the levels above {max_level}")
        return self._expand_calls(functions, top, set())

    def _functions_of(self, solver_program: str):
        """{level: (window min, op program)} of the cycle functions concatenated in `solver_program`."""
        functions = {}
        rest = solver_program or ""
        marker = f"Function {self._cycle_name}@"
        while rest.strip():
            start = rest.find(marker)
            if start < 0:
                raise lowering.UnsupportedExpression("solver_program holds text that is not a cycle function")
            end = rest.find("}\n\n", start)
            text = rest[start:end + 3] if end >= 0 else rest[start:]
            known = self._cycle_functions.get(text)
            if known is None:
                raise lowering.UnsupportedExpression("solver_program holds a cycle function this generator did not produce")
            level, window_min, ops = known
            functions[level] = (window_min, ops)
            rest = rest[start + len(text):]
        return functions

    def _expand_calls(self, functions, level, active):
        """Body of gen_mgCycle@level with every `gen_mgCycle@l()` below it expanded: an evolved function where one is
        defined, else the template's generated cycle from l down (reference_solver.py); on the problem's coarsest level
        the call IS the coarse-grid solver."""
        from evostencils_amd import reference_solver
        coarsest = self._problem.min_level
        if level in active:
            raise lowering.UnsupportedExpression(f"cycle function of level {level} calls itself")
        if level not in functions:
            return reference_solver.reference_solver_ops(self._problem, top_level=level)
        _window_min, ops = functions[level]
        out = []
        for op in ops:
            if op.opcode == abi.OP_COPY and op.dst == abi.F_CGS_RHS and op.level != coarsest:
                # `gen_rhs_u@l = RHS_u@l` in front of a call above the coarsest level: the callee reads its right-hand
                # side from the slot the caller restricted into (evolved callees use RHS_u@l anyway,
                # exastencils.py:303-307; for the template's cycle the two arrays are aliased here)
                continue
            if op.opcode == abi.OP_COARSE_SOLVE and op.level != coarsest:
                if op.level >= level:
                    raise lowering.UnsupportedExpression("a cycle function may only call coarser levels")
                out += self._expand_calls(functions, op.level, active | {level})
                continue
            out.append(op)
        return out

    def _verify_operators(self, stmts):
        """The op program carries no coefficients: make sure the DAG's operator IS the problem's."""
        params = tuple(sorted((k, complex(v)) for k, v in self._problem.parameters.items()))
        for st in stmts:
            if st.kind not in ("smooth", "residual", "richardson"):
                continue
            if st.unknowns or st.operator_row is not None:
                self._verify_system_operator(st)
                continue
            # the operator terminals of the grammar are long-lived objects shared by all candidates: verify each
            # (object, level, PDE parameters) once; the cache keeps the object alive so that its id stays unique
            source = st.operator_entry if st.operator_entry is not None else st.operator_stencil
            key = (id(source), st.level, params)
            if source is not None and self._verified.get(key) is source:
                continue
            stencil = st.operator_stencil
            if stencil is None and st.operator_entry is not None and hasattr(st.operator_entry, "generate_stencil"):
                stencil = st.operator_entry.generate_stencil()
            if stencil is None or hasattr(stencil, "constant_stencils"):
                raise lowering.UnsupportedExpression("system operator must be a constant stencil")
            got = {tuple(o): complex(v) for o, v in stencil.entries}
            if self._problem.system is not None:
                # a scalar problem with a compact nine-point stencil (include/evomg.h: EVOMG_STENCIL_COMPACT): it runs on the
                # systems' path as a system of one unknown
                if len(self._problem.system.field_names) != 1:
                    raise lowering.UnsupportedExpression("the candidate treats a system of PDEs as a scalar problem")
                offsets = list(problems.NINE_POINT_OFFSETS)
                expected = self._problem.system.assembled(st.level)[0, 0]
            else:
                expected = self._problem.mg_stencil(st.level, self._problem.parameters)
                offsets = problems._star_offsets(self.dimension)
            if set(got) - set(offsets):
                raise lowering.UnsupportedExpression("only 5-/7-point star stencils (in 2D: stencils of the 3 x 3 neighbourhood) are supported")
            for o, v in zip(offsets, expected):
                if got.get(o, 0j) != complex(v):
                    raise lowering.UnsupportedExpression(
                        f"operator {st.operator_name}@{st.level} differs from the problem's operator")
            if source is not None:
                self._verified[key] = source

    @staticmethod
    def _constant_stencil(stencil):
        """The IR's stencil algebra (sums of scaled stencils) returns period-1 "multiple" stencils: unwrap them."""
        from evostencils_amd.stencils import multiple
        if hasattr(stencil, "constant_stencils"):
            if any(p != 1 for p in multiple.period_shape(stencil)):
                raise lowering.UnsupportedExpression("system operator entries must be constant-coefficient stencils")
            return multiple.at(stencil, (0,) * stencil.dimension)
        return stencil

    def _verify_system_operator(self, st):
        """Systems of PDEs: the block operator the DAG carries, assembled entry by entry, IS the problem's."""
        import numpy as np
        if self._problem.system is None:
            raise lowering.UnsupportedExpression("the candidate couples several unknowns but the problem has one")
        # as on the scalar path: one comparison per (operator object, level, PDE parameters); the cache keeps the object
        # alive so that its id stays unique. Smoothers carry the whole block operator, residual / Richardson statements
        # one row of entry objects — both are grammar terminals shared by all candidates
        params = tuple(sorted((k, complex(v)) for k, v in self._problem.parameters.items()))
        if st.unknowns:
            source = st.operator_source
            key = ("system", id(source), st.level, params)
        else:
            source = tuple(st.operator_row)
            key = ("system-row", st.dst.index, tuple(id(e) for e in source), st.level, params)
        cached = self._verified.get(key)
        if source is not None and cached is not None and (cached is source or cached == source):
            return
        expected = self._problem.system.assembled(st.level)
        rows = st.system_stencils if st.unknowns else None
        if rows is None:
            i = st.dst.index
            rows = {i: [self._constant_stencil(e.generate_stencil()) for e in st.operator_row]}
            items = rows.items()
        else:
            items = enumerate(rows)
        for i, row in items:
            for j, stencil in enumerate(row):
                got = np.zeros(9)
                for off, val in stencil.entries:
                    if tuple(off) not in problems.NINE_POINT_OFFSETS:
                        raise lowering.UnsupportedExpression("system operators wider than 9 points are not supported")
                    got[problems.NINE_POINT_OFFSETS.index(tuple(off))] += float(val)
                if not np.array_equal(got, expected[i, j]):
                    raise lowering.UnsupportedExpression(f"operator entry ({i}, {j})@{st.level} differs from the problem's")
        if source is not None:
            self._verified[key] = source

    # ---- THE call (exastencils.py:485-537) -----------------------------------------------------------
    def generate_and_evaluate(self, expression, storages, min_level: int, max_level: int, solver_program: str,
                              infinity=1e100, evaluation_samples=3, global_variable_values=None):
        if global_variable_values is None:
            global_variable_values = {}
        try:
            ops = self.generate_op_program(expression, min_level, max_level, solver_program)
        except lowering.UnsupportedExpression:
            return infinity, infinity, infinity
        return self.evaluate_op_program(ops, min_level, max_level, solver_program, infinity, evaluation_samples,
                                        global_variable_values)

    def evaluate_op_program(self, ops, min_level: int, max_level: int, solver_program: str = "", infinity=1e100,
                            evaluation_samples=3, global_variable_values=None):
        """The second half of generate_and_evaluate: run an op program `generate_op_program` returned (the part that
        releases the interpreter lock; ProgramGeneratorPool lowers under a lock and evaluates outside it)."""
        if global_variable_values is None:
            global_variable_values = {}
        if solver_program or max_level < self._problem.max_level or min_level > self._problem.min_level:
            # the executable always solves the problem on its finest level with the whole hierarchy underneath
            min_level, max_level = self._problem.min_level, self._problem.max_level
        self._counter += 1
        mapping = dict(global_variable_values)
        n = 3 if "k" in mapping else 1          # Helmholtz: k, 2k, 4k (exastencils.py:518-532)
        total_t = total_rho = total_it = 0
        for _ in range(n):
            runtime, rho, iterations = self.compile_and_run(ops, min_level, max_level, mapping=mapping,
                                                            infinity=infinity,
                                                            evaluation_samples=evaluation_samples)
            total_t += runtime
            total_rho += rho
            total_it += iterations
            if iterations >= infinity or rho > 1:
                return total_t, total_rho, total_it
            if n > 1:
                mapping["k"] *= 2
        return total_t / n, total_rho / n, total_it / n

    def coarse_iteration_cost_ms(self, min_level: int, max_level: int):
        """What one iteration of the coarse-grid solver costs on this level range, as far as `isolated_time_ms` has settled
        it (None: not yet)."""
        state = self._coarse_iteration_ms.get((min_level, max_level))
        return None if state is None else state["known"]

    def isolated_time_ms(self, ops, min_level: int, max_level: int, iterations: float, solver_program: str = "",
                         global_variable_values=None, coarse_iterations=None, probe_iterations: int = 6):
        """Time-to-solution of a candidate as it would be ALONE on the GPU, for an evaluation that ran with other candidates
        in flight (where the measured wall time includes the neighbours and would change the selection pressure of the time
        objective, optimization/program.py:449-451). The caller makes sure nothing else runs on the device meanwhile. Three
        short probes (evomg_time_model) give the cost of what a solve pays once, of one solver-loop iteration without its
        coarse-grid Krylov iterations, and of one such Krylov iteration; `iterations` and `coarse_iterations` of the real
        evaluation (the latter: `last_coarse_iterations` of the generator that ran it) scale them. The Krylov count has to
        be part of the model: on the headline sizes the coarse solve is most of an iteration and its iteration count drifts
        by tens of per cent along a solve and between candidates (profiles/r04_time_objective.md)."""
        if solver_program or max_level < self._problem.max_level or min_level > self._problem.min_level:
            min_level, max_level = self._problem.min_level, self._problem.max_level
        handle = self._ensure_handle(min_level, max_level, dict(global_variable_values or {}))
        its = float(iterations)
        n_long = max(2, min(int(probe_iterations), int(its)))
        if n_long < 3 or coarse_iterations is None:
            ms, _done = handle.time_iterations(ops, max(1, min(int(its), max(n_long, 1))))
            return ms * its
        # what ONE iteration of the coarse-grid solver costs belongs to the problem and its coarsest level, not to the candidate:
        # the first few candidates measure it (three probes each), every later one runs the two cheap probes only
        key = (min_level, max_level)
        state = self._coarse_iteration_ms.setdefault(key, {"samples": [], "known": None, "uses": 0})
        known = state["known"]
        if known is not None and state["uses"] < self.COARSE_COST_REFRESH:
            state["uses"] += 1
            fixed, per_iteration, per_coarse = handle.time_model(ops, max(1, n_long // 3), n_long, per_coarse=known)
        else:
            # not known yet, or due for another look: clocks and what shares the device drift over a long optimisation
            fixed, per_iteration, per_coarse = handle.time_model(ops, max(1, n_long // 3), n_long)
            samples = state["samples"]
            samples.append(per_coarse)
            positive = sorted(v for v in samples[-5:] if v > 0.0)
            if len(positive) >= (self.COARSE_COST_SAMPLES if known is None else 1):
                state["known"], state["uses"] = float(positive[len(positive) // 2]), 0
            elif len(samples) >= 4 * self.COARSE_COST_SAMPLES:
                state["known"], state["uses"] = 0.0, 0      # a coarse solver that does not iterate: nothing to scale
            if known is not None:
                per_coarse = state["known"]
        return max(fixed, 0.0) + per_iteration * its + per_coarse * float(coarse_iterations)

    def compile_and_run(self, ops, min_level, max_level, mapping=None, infinity=1e100, evaluation_samples=3):
        """exastencils.py:468-483 + evaluate :417-443; nothing is compiled here.

        The `(infinity,)*3` sentinel is the reference's answer to ONE bad candidate (a failed compile or run of its
        executable): a program the backend rejects (ERR_INVALID / ERR_UNSUPPORTED) or a solve that exceeds the
        evaluation timeout. A failing DEVICE (no GPU, out of memory, launch failure, a coarse-grid rendezvous that
        timed out twice) is not a property of the candidate: it raises, so that a sick GPU or rank stops the run
        instead of silently turning whole generations into 'bad candidates'."""
        handle = self._ensure_handle(min_level, max_level, mapping)      # device errors propagate
        cap = self._problem.max_iterations + 2
        rc, out = handle.try_evaluate(ops, evaluation_samples, 1e3 * float(self.timeout_evaluate), cap)
        if rc == abi.ERR_DEVICE:
            raise abi.BackendError(f"evomg_evaluate failed on device {self._device_id}: {handle.last_error()}")
        self.last_coarse_iterations = handle.coarse_iterations() if rc == abi.OK else None
        if rc != abi.OK or out.status == abi.SOLVE_WALL_LIMIT:
            return infinity, infinity, infinity
        t, rho, its = self.interpret_outcome(out, infinity, self._solver_iteration_limit,
                                             self._problem.outer_kind == abi.OUTER_PBICGSTAB)
        if its >= infinity or rho > 1 or rho >= infinity:
            return t, rho, its             # evaluate() returns such a sample as it is (exastencils.py:436-439)
        # evaluate() sums the factors and counts of the samples and divides by their number (exastencils.py:440-443).
        # The solves are deterministic, every sample contributes the same values; what is left of the averaging is
        # the rounding of (a + a + a) / 3, visible in the last digit of notebooks/tutorial.ipynb:3373.
        sum_rho = sum_its = 0
        for _ in range(evaluation_samples):
            sum_rho += rho
            sum_its += its
        return t, sum_rho / evaluation_samples, sum_its / evaluation_samples

    @staticmethod
    def _as_printed(value):
        """The reference reads every number back from the executable's stdout (exastencils.py:539-577); the generated
        C++ prints doubles with the stream default of 6 significant digits. Rounding each per-iteration factor the same
        way is what makes notebooks/tutorial.ipynb:3373 reproducible to its last digit."""
        return float("%.6g" % value)

    @staticmethod
    def interpret_outcome(out: abi.SolveOutcome, infinity, solver_iteration_limit=None, krylov_outer=False):
        """What `parse_output` + `evaluate` derive from the executable's stdout (exastencils.py:417-443,
        539-584), computed from the residual history instead of printed text."""
        rho_inf = math.sqrt(infinity)
        hist = out.history
        factors = []
        count = 0
        if krylov_outer:
            # the Helmholtz solver prints ONE line with |res|/|initRes| (Helmholtz exa3:193,198)
            pairs = [(hist[0], hist[-1])] if hist is not None and len(hist) >= 2 else []
        else:
            pairs = list(zip(hist[:-1], hist[1:])) if hist is not None else []
        for prev, cur in pairs:
            rho = cur / prev if prev != 0 else float("nan")
            if math.isinf(rho) or math.isnan(rho):
                factors.append(rho_inf)
            else:
                factors.append(ProgramGenerator._as_printed(rho))
                count += 1
        if count > 0:
            convergence_factor = 1
            exponent = 1.0 / len(factors)
            for rho in factors:
                convergence_factor *= math.pow(rho, exponent)
        else:
            convergence_factor = infinity
        iterations = out.n_iterations
        if krylov_outer:
            iterations -= 1      # `repeat ... count curStep` prints the 0-based step (exa3:160,193)
            if out.status != abi.SOLVE_CONVERGED:
                # the hand-written solver prints "Maximum number of solver iterations (...) was exceeded"
                # (Helmholtz exa3:198), which parse_output maps to infinity (exastencils.py:555-557)
                iterations = infinity
        elif out.status == abi.SOLVE_DIVERGED:
            # the generated loop keeps printing nan factors up to the cap; the candidate is discarded through its
            # convergence factor (> 1), the count is the cap
            pass
        # The ExaStencils-GENERATED solver loop prints no such message: at the cap the last "Residual after <n>
        # iterations" line simply carries n = solver_maxNumIts, a finite count (notebooks/tutorial.ipynb:3373:
        # "number of iterations: 100.0" with solver_maxNumIts = 100, solver_iteration_limit = 500).
        if iterations == 0:
            iterations = infinity
        if solver_iteration_limit is not None and iterations >= solver_iteration_limit:
            iterations = infinity
        time_to_solution = out.time_ms
        if iterations >= infinity or convergence_factor > 1:
            return time_to_solution, convergence_factor, iterations
        if math.isinf(convergence_factor) or math.isnan(convergence_factor):
            return infinity, infinity, infinity
        return time_to_solution, convergence_factor, iterations

    # ---- pieces of the reference interface that have no work left to do ---------------------------------
    def invalidate_solver_cache(self):
        self._solver_cache = {}

    def close(self):
        if self._handle is not None:
            self._handle.close()
            self._handle = None


class ProgramGeneratorPool:
    """Several candidates in flight on ONE GPU behind the ProgramGenerator interface: `n_workers` generators (own HIP
    stream and own set of level fields each) and a thread-safe `generate_and_evaluate` that hands every call to a free
    one. On the reference side this is two lines — the pool instead of the generator, and a thread pool as the toolbox's
    `map` (the optimizer evaluates a population with `toolbox.map(toolbox.evaluate, ...)`, optimization/program.py:491,577):

        program_generator = ProgramGeneratorPool(lambda: ProgramGenerator(compiler_path, base_path, ...), 12)
        optimizer.toolbox.register("map", multiprocessing.pool.ThreadPool(12).map)          # after Optimizer(...)

    Why it pays: a candidate's coarse-grid Krylov solve is a 1 ms latency chain on one of 256 compute units; the sweeps of
    the other candidates use the rest of the GPU meanwhile (5.5x on BASELINE's 256-candidate population at 2048^2,
    profiles/r03_throughput_queues.md). Convergence factors and iteration counts are those of the sequential evaluation;
    a candidate's measured time includes its neighbours unless time_objective="isolated" is asked for (below: short probes on
    the idle GPU give every candidate the time a sequential evaluation would have measured, at 41.7 instead of 46.7 solves/s).

    Everything that is not an evaluation — attributes, generate_storage, generate_cycle_function ... — is answered by the
    first generator; level-range set-up calls go to all of them; the cycle functions of earlier optimisation runs
    (`solver_program`) are known to all (one shared table). Lowering mutates flags on the DAG, whose terminals all
    candidates share: it runs under one lock (it holds the interpreter lock anyway); the solve itself does not."""

    def __init__(self, make_generator, n_workers: int, time_objective: str = "contended", probe_iterations: int = 6):
        """time_objective: "contended" — a candidate's time is the wall time of its solve, neighbours included (fastest);
        "isolated" — every evaluation is followed by a short probe of the same candidate while NOTHING else runs on the
        GPU (ProgramGenerator.isolated_time_ms): the workers finish the solves they have in flight, one of them runs the
        pending probes one after the other, then all go on. rho and iteration counts are the same either way; with
        "isolated" the time objective is the one a sequential evaluation would have measured, so the pool can stand behind
        the optimizer without changing its selection (optimization/program.py:449-451)."""
        import queue
        import threading
        if n_workers < 1:
            raise ValueError("n_workers must be at least 1")
        if time_objective not in ("contended", "isolated"):
            raise ValueError("time_objective must be 'contended' or 'isolated'")
        from .. import population
        population.warn_if_too_many_workers(n_workers)
        self._time_objective, self._probe_iterations = time_objective, probe_iterations
        self._cond = threading.Condition()
        self._active = 0                # solves in flight
        self._probing = False           # a probe batch owns the GPU: no new solve starts
        self._pending = []              # probe requests of workers whose solve is done
        self._generators = [make_generator() for _ in range(n_workers)]
        shared = self._generators[0]._cycle_functions
        for g in self._generators[1:]:
            g._cycle_functions = shared
        self._free = queue.Queue()
        for g in self._generators:
            self._free.put(g)
        self._lower_lock = threading.Lock()

    def __getattr__(self, name):            # only reached for names the pool itself does not define
        return getattr(self._generators[0], name)

    def __setattr__(self, name, value):
        if name in ("_counter", "_average_generation_time"):        # reset by the optimizer (program.py:850-851)
            for g in self._generators:
                setattr(g, name, value)
        else:
            object.__setattr__(self, name, value)

    @property
    def generators(self):
        return list(self._generators)

    def initialize_code_generation(self, min_level: int, max_level: int):
        for g in self._generators:
            g.initialize_code_generation(min_level, max_level)

    def reinitialize(self, min_level, max_level, global_expressions):
        for g in self._generators:
            g.reinitialize(min_level, max_level, global_expressions)

    def generate_and_evaluate(self, expression, storages, min_level: int, max_level: int, solver_program: str,
                              infinity=1e100, evaluation_samples=3, global_variable_values=None):
        g = self._free.get()
        try:
            with self._lower_lock:
                try:
                    ops = g.generate_op_program(expression, min_level, max_level, solver_program)
                except lowering.UnsupportedExpression:
                    return infinity, infinity, infinity
            if self._time_objective != "isolated":
                return g.evaluate_op_program(ops, min_level, max_level, solver_program, infinity, evaluation_samples,
                                             global_variable_values)
            with self._cond:
                while self._probing:
                    self._cond.wait()
                self._active += 1
            try:
                t, rho, its = g.evaluate_op_program(ops, min_level, max_level, solver_program, infinity, evaluation_samples,
                                                    global_variable_values)
            except BaseException:
                # this solve may have been the last one in flight while other workers are parked behind their probe
                # requests: somebody has to run those probes — the same hand-over as in _probe, then the error goes up
                self._leave(None)
                raise
            return self._probe(g, ops, min_level, max_level, solver_program, global_variable_values, t, rho, its, infinity)
        finally:
            self._free.put(g)

    def _probe(self, g, ops, min_level, max_level, solver_program, global_variable_values, t, rho, its, infinity):
        """Second half of an evaluation under time_objective="isolated": queue this candidate's probe, and — if this worker's
        solve was the last one in flight — run every queued probe alone on the GPU; otherwise wait for the worker that does."""
        request = None
        if t < infinity and its < infinity and its >= 1:
            request = {"g": g, "args": (ops, min_level, max_level, its, solver_program, global_variable_values,
                                        getattr(g, "last_coarse_iterations", None), self._probe_iterations),
                       "done": False, "time": t, "error": None}
        self._leave(request)
        if request is None:
            return t, rho, its
        with self._cond:
            while not request["done"]:
                self._cond.wait()
        if request["error"] is not None:
            raise request["error"]
        return request["time"], rho, its

    def _leave(self, request):
        """A worker's solve is over (with a probe request, without one, or with an exception): one solve less in flight; the
        worker whose solve was the LAST one in flight runs every queued probe, alone on the GPU, and wakes the others."""
        with self._cond:
            self._active -= 1
            if request is not None:
                self._pending.append(request)
            if self._active == 0 and self._pending and not self._probing:
                self._probing = True
                batch, self._pending = self._pending, []
            else:
                batch = None
                self._cond.notify_all()
        if batch is None:
            return
        try:
            for r in batch:                     # nothing else is in flight: the wall time of the probe is the candidate's own
                try:
                    r["time"] = r["g"].isolated_time_ms(*r["args"])
                except Exception as exc:        # noqa: BLE001 — handed to the worker that owns the request
                    r["error"] = exc
        finally:
            with self._cond:
                for r in batch:
                    r["done"] = True            # (a request a BaseException cut short keeps its contended time)
                self._probing = False
                self._cond.notify_all()

    def close(self):
        for g in self._generators:
            g.close()
