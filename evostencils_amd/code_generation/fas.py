"""Drop-in counterpart of the reference's second backend, the non-linear full-approximation-scheme generator
``evostencils.code_generation.exastencils_FAS.ProgramGeneratorFAS``.

The reference lowers a FAS cycle DAG to ExaSlang *layer 4* statements (exastencils_FAS.py:99-319 on the AST of
layer4.py), appends them to the problem's ``*_template.exa4``, runs the ExaStencils JVM, make and ``likwid-pin
./exastencils``, and parses residual norms out of stdout (:370-394). Here ONE walk over the DAG yields both

* the layer-4 text of the cycle function — character for character what the reference prints (golden tests against
  text produced by importing the reference, tests/golden/generate_golden.py) — and
* the flat op program of the FAS vocabulary (include/evomg.h: EVOMG_OP_FAS_*), which the HIP backend executes.

What the template fixes (parsed from the ``.exa4`` template, nothing is hard-wired): the non-linear stencil
``gam * exp(Solution)``, the value of ``gam``, the coarse-grid solver (``repeat 200 times`` the Newton-Jacobi
``Smoother`` with weight 0.8), the stopping rule of ``Solve`` (1e-10 reduction or 300 cycles) and the right-hand side
function. Numerics: no output of a reference FAS run is recorded anywhere in the reference, so — unlike the linear
path — the oracle for these ops is pinned by the template's own formulas only ("parity unpinned").
"""
import math
import os
import re
from dataclasses import dataclass
from statistics import mean

import numpy as np
import sympy
from sympy.parsing.sympy_parser import parse_expr

from evostencils_amd import abi
from evostencils_amd.grammar import multigrid as initialization
from evostencils_amd.ir import base


class UnsupportedFASExpression(Exception):
    """The DAG contains a shape the FAS grammar does not produce."""


# ---- what the walk returns for a sub-expression: its layer-4 text and what it denotes -------------------------------
@dataclass
class _Term:
    text: str
    kind: str = "expr"      # field | stencil | prolongated | restricted_residual | operator_of_approximation | coarse_rhs | expr
    name: str = ""
    level: int = -1


def _field(name, level):
    return _Term(f"{name}@{level}", "field", name, level)


def _stencil(name, level):
    return _Term(f"{name}@{level}", "stencil", name, level)


# ---- the problem template ------------------------------------------------------------------------------------------
@dataclass
class FASTemplate:
    dimension: int
    min_level: int
    max_level: int
    gamma: float
    nonlinear_text: str             # stencil entry, e.g. "( gam * exp( Solution<active>@current ) )"
    cgs_sweeps: int
    cgs_omega: float
    target_res_reduction: float
    max_iterations: int
    rhs_function: object            # callable (x, y) -> array
    solution_function: object


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def _function_body(text, name):
    m = re.search(r"Function\s+" + re.escape(name) + r"(?:@\S+)?\s*\([^)]*\)\s*(?::\s*\w+)?\s*\{", text)
    if not m:
        raise ValueError(f"template: Function {name} not found")
    depth, i = 1, m.end()
    while depth and i < len(text):
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.end():i - 1]


def _exaslang_function(text, name, constants, others=()):
    """(xPos, yPos) -> numpy function from `Function name ( xPos : Real, yPos : Real ) : Real { return ( ... ) }`."""
    body = _function_body(text, name)
    expr = body[body.index("return") + len("return"):].strip()
    x, y = sympy.symbols("xPos yPos")
    local = {"xPos": x, "yPos": y, "PI": sympy.pi, "exp": sympy.exp, "sin": sympy.sin, "cos": sympy.cos}
    local.update(constants)
    for other_name, other in others:
        local[other_name] = other
    parsed = parse_expr(expr.replace("^", "**"), local_dict=local)
    return sympy.lambdify((x, y), parsed, "numpy"), (lambda a, b, e=parsed: e.subs({x: a, y: b}))


# ---- the non-linear stencil entry: sympy expression -> reverse-Polish program (include/evomg.h: evomg_nl_program_t) -------
SOLUTION_SYMBOL = sympy.Symbol("Solution")
_NL_FUNCTIONS = {sympy.exp: abi.NL_EXP, sympy.sin: abi.NL_SIN, sympy.cos: abi.NL_COS, sympy.log: abi.NL_LOG, sympy.tanh: abi.NL_TANH,
                 sympy.sinh: abi.NL_SINH, sympy.cosh: abi.NL_COSH}


def nonlinear_expression(nonlinear_text, solution="Solution"):
    """The stencil entry as a sympy expression of `Solution`, slot and level modifiers removed exactly as the reference does
    (parser.extract_nonlinear_term, parser.py:146-169)."""
    expr = re.sub("<[^>]+>", "", nonlinear_text)
    expr = re.sub("@[^)]+", "", expr)
    try:
        return parse_expr(expr.replace("^", "**"), local_dict={solution: SOLUTION_SYMBOL, "gam": sympy.Symbol("gam")})
    except Exception as exc:        # noqa: BLE001 — whatever sympy makes of a malformed entry: the problem is not ours to solve
        raise UnsupportedFASExpression(f"non-linear stencil entry `{nonlinear_text}`: {type(exc).__name__}: {exc}") from exc


def _rpn(expr, consts, values):
    """Opcodes of `expr` in evaluation order (post-order, arguments left to right in sympy's canonical order)."""
    if expr == SOLUTION_SYMBOL:
        return [abi.NL_PUSH_U]
    if expr.is_Symbol or expr.is_Number or expr.is_NumberSymbol:
        if expr.is_Symbol:
            if str(expr) not in values:
                raise UnsupportedFASExpression(f"non-linear stencil entry: unknown symbol {expr}")
            v = float(values[str(expr)])
        else:
            v = float(expr)
        if v not in consts:
            if len(consts) >= abi.NL_MAX_CONSTS:
                raise UnsupportedFASExpression("non-linear stencil entry: too many constants")
            consts.append(v)
        return [abi.NL_PUSH_CONST + consts.index(v)]
    if expr.is_Add or expr.is_Mul:
        args = list(expr.args)
        if expr.is_Mul and args[0] == -1 and len(args) >= 2:          # -x, -x*y: negate the product of the rest
            return _rpn(sympy.Mul(*args[1:]), consts, values) + [abi.NL_NEG]
        code = _rpn(args[0], consts, values)
        for a in args[1:]:
            if expr.is_Add and a.is_Mul and a.args[0] == -1:          # a - b
                code += _rpn(sympy.Mul(*a.args[1:]), consts, values) + [abi.NL_SUB]
            else:
                code += _rpn(a, consts, values) + [abi.NL_ADD if expr.is_Add else abi.NL_MUL]
        return code
    if expr.is_Pow:
        base_, e = expr.args
        if e == sympy.Rational(1, 2):
            return _rpn(base_, consts, values) + [abi.NL_SQRT]
        if e == sympy.Rational(-1, 2):
            return _rpn(sympy.Integer(1), consts, values) + _rpn(base_, consts, values) + [abi.NL_SQRT, abi.NL_DIV]
        if e.is_Integer and 2 <= int(e) <= 8:
            return _rpn(base_, consts, values) + [abi.NL_DUP] * (int(e) - 1) + [abi.NL_MUL] * (int(e) - 1)
        if e.is_Integer and -8 <= int(e) <= -1:
            return _rpn(sympy.Integer(1), consts, values) + _rpn(sympy.Pow(base_, -e), consts, values) + [abi.NL_DIV]
        raise UnsupportedFASExpression(f"non-linear stencil entry: power {e} is not supported")
    op = _NL_FUNCTIONS.get(expr.func)
    if op is not None and len(expr.args) == 1:
        return _rpn(expr.args[0], consts, values) + [op]
    raise UnsupportedFASExpression(f"non-linear stencil entry: {expr.func} is not supported")


def _stack_depth(code):
    depth = peak = 0
    for op in code:
        if op == abi.NL_PUSH_U or op == abi.NL_DUP or abi.NL_PUSH_CONST <= op < abi.NL_PUSH_CONST + abi.NL_MAX_CONSTS:
            depth += 1
        elif abi.NL_ADD <= op <= abi.NL_DIV:
            depth -= 1
        peak = max(peak, depth)
    return peak


def compile_nonlinear(sigma_expr, values):
    """(sigma opcodes, Newton-term opcodes, constants) of a stencil entry sigma(Solution): the entry itself and
    d/dSolution [sigma * Solution], simplified by sympy as the reference's generator forms it (exastencils_FAS.py: sympy_jacobian).
    `values`: the template's `Val` constants (gam). Raises UnsupportedFASExpression for what the programs cannot express."""
    jacobian_expr = sympy.simplify(sympy.diff(sigma_expr * SOLUTION_SYMBOL, SOLUTION_SYMBOL))
    consts = []
    sigma = _rpn(sigma_expr, consts, values)
    jacobian = _rpn(jacobian_expr, consts, values)
    for code in (sigma, jacobian):
        if len(code) > abi.NL_MAX_OPS or _stack_depth(code) > abi.NL_MAX_STACK:
            raise UnsupportedFASExpression("non-linear stencil entry: too long or too deep for a straight-line program")
    return sigma, jacobian, consts


def evaluate_program(code, consts, u):
    """Host-side evaluation of a program (tests, and the cross-check of classify_nonlinear): the same stack machine."""
    stack = []
    for op in code:
        if op == abi.NL_PUSH_U:
            stack.append(float(u))
        elif op == abi.NL_DUP:
            stack.append(stack[-1])
        elif abi.NL_PUSH_CONST <= op < abi.NL_PUSH_CONST + abi.NL_MAX_CONSTS:
            stack.append(consts[op - abi.NL_PUSH_CONST])
        elif abi.NL_ADD <= op <= abi.NL_DIV:
            b, a = stack.pop(), stack.pop()
            stack.append(a + b if op == abi.NL_ADD else a - b if op == abi.NL_SUB else a * b if op == abi.NL_MUL else a / b)
        else:
            f = {abi.NL_NEG: lambda v: -v, abi.NL_EXP: math.exp, abi.NL_SIN: math.sin, abi.NL_COS: math.cos,
                 abi.NL_SQRT: math.sqrt, abi.NL_LOG: math.log, abi.NL_TANH: math.tanh, abi.NL_SINH: math.sinh,
                 abi.NL_COSH: math.cosh}[op]
            stack[-1] = f(stack[-1])
    assert len(stack) == 1
    return stack[0]


def classify_nonlinear(nonlinear_text, gamma):
    """(abi.NONLINEAR_* kind, program or None) of a template's stencil entry. The reference's example — gam * exp(Solution) —
    keeps its dedicated kernels (kind GAM_U_EXP_U: one exponential serves the entry and the Newton term); every other entry
    becomes a pair of programs."""
    sigma = nonlinear_expression(nonlinear_text)
    if sympy.simplify(sigma - sympy.Symbol("gam") * sympy.exp(SOLUTION_SYMBOL)) == 0:
        return abi.NONLINEAR_GAM_U_EXP_U, None
    return abi.NONLINEAR_PROGRAM, compile_nonlinear(sigma, {"gam": gamma})


def read_template(build_path, knowledge_file, template_file, nonlinear_stencil):
    dimension = min_level = max_level = None
    with open(os.path.join(build_path, knowledge_file)) as fh:       # parser.extract_knowledge_information, parser.py:114-125
        for line in fh:
            tokens = line.split("=")
            lhs = tokens[0].strip(" \n\t")
            if lhs == "dimensionality":
                dimension = int(tokens[1].strip(" \n\t"))
            elif lhs == "minLevel":
                min_level = int(tokens[1].strip(" \n\t"))
            elif lhs == "maxLevel":
                max_level = int(tokens[1].strip(" \n\t"))
    with open(os.path.join(build_path, template_file)) as fh:
        raw = fh.read()
    text = _strip_comments(raw)
    m = re.search(r"Stencil\s+" + re.escape(nonlinear_stencil) + r"@\w+\s*\{[^=]*=>\s*([^\n]*)\n", text)
    if not m:
        raise ValueError(f"template: Stencil {nonlinear_stencil} not found")
    nonlinear_text = m.group(1).strip()
    gam = re.search(r"Val\s+gam\s*:\s*Real\s*=\s*([-+0-9.eE]+)", text)
    gamma = float(gam.group(1)) if gam else 0.0
    cgs = _function_body(text, "CGS")
    sweeps = re.search(r"repeat\s+(\d+)\s+times", cgs)
    smoother = _function_body(text, "Smoother")
    weight = re.search(r"\+\s*([0-9.]+)\s*\*\s*\(", smoother)
    solve = _function_body(text, "Solve")
    stop = re.search(r"repeat\s+until\s*\(\s*res_0\s*<\s*\(\s*([-+0-9.eE]+)\s*\*\s*resStart_0\s*\)\s*\|\|\s*numIt\s*>=\s*(\d+)", solve)
    constants = {"gam": gamma}
    sol_np, sol_sym = _exaslang_function(text, "solFct", constants)
    rhs_np, _ = _exaslang_function(text, "rhsFct", constants, others=[("solFct", sol_sym)])
    return FASTemplate(dimension, min_level, max_level, gamma, nonlinear_text,
                       int(sweeps.group(1)) if sweeps else 200, float(weight.group(1)) if weight else 0.8,
                       float(stop.group(1)) if stop else 1e-10, int(stop.group(2)) if stop else 300, rhs_np, sol_np)


class ProgramGeneratorFAS:
    """Same constructor arguments, attributes and methods as the reference class (exastencils_FAS.py:11-96,
    396-445); `library` / `device_id` select the HIP backend (no ExaStencils run, `exastencils_compiler` and
    `platform_file` are accepted and ignored); `levels` = (min, max) overrides the .knowledge file's level range."""

    def __init__(self, problem_name, solution, rhs, residual, FASApproximation, restriction, prolongation, op_linear,
                 op_nonlinear, fct_name_mgcycle, fct_cgs, fct_smoother=None, mpi_rank=0, platform_file=None,
                 build_path=None, exastencils_compiler=None, library=None, device_id=None, levels=None):
        self.problem_name = problem_name
        self.platform_file, self.build_path, self.exastencils_compiler = platform_file, build_path, exastencils_compiler
        self.knowledge_file = os.path.join(problem_name, f"{problem_name}.knowledge")
        self.exa_file_template = os.path.join(problem_name, f"{problem_name}_template.exa4")
        self.mpi_rank = mpi_rank
        self.fct_CGS, self.fct_smoother = fct_cgs, fct_smoother
        self._names = dict(solution=solution, rhs=rhs, residual=residual, approximation=FASApproximation,
                           restriction=restriction, prolongation=prolongation, op_linear=op_linear,
                           op_nonlinear=op_nonlinear)
        if op_nonlinear is None:
            raise UnsupportedFASExpression("a FAS problem without a non-linear stencil is the linear backend's job")
        self.template = read_template(build_path, self.knowledge_file, self.exa_file_template, op_nonlinear)
        if levels is not None:          # not in the reference: another level range than the .knowledge file's (tests, benchmarks)
            import dataclasses
            self.template = dataclasses.replace(self.template, min_level=levels[0], max_level=levels[1])
        self.dimension, self.min_level, self.max_level = self.template.dimension, self.template.min_level, self.template.max_level
        # derivative of the non-linear term as sympy prints it (parser.extract_nonlinear_term, parser.py:146-169)
        expr = re.sub("<[^>]+>", "", self.template.nonlinear_text)
        expr = re.sub("@[^)]+", "", expr)
        self.sympy_expr_nonlinear = parse_expr(expr + " * " + solution)
        self.sympy_symbol_solution = sympy.Symbol(solution)
        self.sympy_jacobian = sympy.simplify(sympy.diff(self.sympy_expr_nonlinear, self.sympy_symbol_solution))
        self.update_rhs = {level: True for level in range(self.min_level, self.max_level + 1)}
        self.relaxation_factors, self.partitioning = [], []
        self.fct_name = fct_name_mgcycle + f"@{self.max_level}"
        self._lines, self._ops, self._ops_ok = [], [], True
        # what the optimiser reads to build its primitive set (exastencils_FAS.py:75-96)
        self.equations, self.operators, self.fields = [], [], [sympy.Symbol("u")]
        for i in range(self.min_level, self.max_level + 1):
            self.equations.append(initialization.EquationInfo("solEq", i, f"( Laplace@{i} * u@{i} ) == RHS_u@{i}"))
            self.operators.append(initialization.OperatorInfo("RestrictionNode", i, None, base.Restriction))
            self.operators.append(initialization.OperatorInfo("ProlongationNode", i, None, base.Prolongation))
            self.operators.append(initialization.OperatorInfo("Laplace", i, None, base.Operator))
        size = 2 ** self.max_level
        h = 1 / (2 ** self.max_level)
        self.coarsening_factor = [tuple([2] * self.dimension) for _ in self.fields]
        self.finest_grid = [base.Grid(tuple([size] * self.dimension), tuple([h] * self.dimension), self.max_level)
                            for _ in self.fields]
        self._library, self._device_id, self._handle = library, device_id, None

    uses_FAS = property(lambda self: True)

    # ---- the walk (exastencils_FAS.py:99-319) ---------------------------------------------------------------------
    def _emit(self, line):
        self._lines.append(line)

    def _op(self, *args, **kw):
        self._ops.append(abi.make_op(*args, **kw))

    def _n(self, key, level):
        return f"{self._names[key]}@{level}"

    def _operator_of(self, operand_text, level):
        """( Laplace@l * X ) [+ ( gamSten@l * X )]"""
        text = f"( {self._n('op_linear', level)} * {operand_text} )"
        if self._names["op_nonlinear"] is not None:
            text = f"( {text} + ( {self._n('op_nonlinear', level)} * {operand_text} ) )"
        return text

    def _update_rhs(self, rhs_obj, level):
        if type(rhs_obj).__name__ == "RightHandSide":
            return
        assert level < self.max_level
        if self.update_rhs[level]:
            term = self.traverse_graph(rhs_obj)
            rhs = self._n("rhs", level)
            self._emit(f"loop over {rhs} {{\n  {rhs} = {term.text}\n}}")
            if term.kind == "coarse_rhs" and term.level == level:
                self._op(abi.OP_FAS_COARSE_RHS, level, abi.F_RHS, abi.F_RES)
            else:
                self._ops_ok = False
            self.update_rhs[level] = False

    def _smoothing(self, expression, level):
        residual = expression.operand2
        self._update_rhs(residual.rhs, level)
        omega = self.relaxation_factors.pop()
        partitioning = self.partitioning.pop()
        red_black = getattr(partitioning, "__name__", type(partitioning).__name__) == "RedBlack"
        solution = self._n("solution", level)
        if self.fct_smoother is not None:
            # a smoother function of the template instead of generated statements (exastencils_FAS.py:266-267); the
            # template's `Smoother` is the Newton-Jacobi sweep of the coarse-grid solver: one local step, its weight
            self._emit(f"{self.fct_smoother}@{level} (  )")
            if self.fct_smoother == "Smoother":
                self._op(abi.OP_FAS_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS, abi.PART_SINGLE, (1, 1, 1), self.template.cgs_omega)
            else:
                self._ops_ok = False
            return
        operator = expression.operand1.operand
        numerator = f"( {self._n('rhs', level)} - {self._operator_of(solution, level)} )"
        diag = f"{self._n('op_linear', level)}:[0,0]"
        n_newton = 0
        if type(operator).__name__ == "Addition":
            jacobian = str(self.sympy_jacobian.subs(self.sympy_symbol_solution, sympy.Symbol(solution)))
            n_newton = operator.operand2.n_newton_steps
            correction = f"( {numerator} / ( {diag} + {jacobian} ) )"
        else:
            correction = f"( {numerator} / {diag} )"
        steps = [f"{solution} += {omega} * {correction}"] * max(n_newton, 1)
        if red_black:
            body = "\n  ".join(steps)
            self._emit("color with {\n  ( i0 + i1 ) % 2,\n  "
                       + f"communicate {solution}\n  loop over {solution} {{\n  {body}\n}}\n  apply bc to {solution}" + "\n}")
        else:
            name, lvl = self._names["solution"], level
            body = "\n  ".join([f"Variable Solution_old : Real = {solution}"] + steps
                               + [f"{name}<next>@{lvl} = {solution}", f"{solution} = Solution_old"])
            self._emit(f"communicate {solution}")
            self._emit(f"loop over {solution} {{\n  {body}\n}}")
            self._emit(f"advance {solution}")
        self._op(abi.OP_FAS_SMOOTH, level, abi.F_SOL, 0, abi.F_RHS,
                 abi.PART_RED_BLACK if red_black else abi.PART_SINGLE, (n_newton, 1, 1), float(omega))

    def traverse_graph(self, expression):
        level = expression.grid[0].level
        kind = type(expression).__name__
        if kind == "Cycle":
            self.relaxation_factors.append(expression.relaxation_factor)
            self.partitioning.append(expression.partitioning)
            solution = self.traverse_graph(expression.approximation)
            correction = self.traverse_graph(expression.correction)
            if correction is not None:
                omega = self.relaxation_factors.pop()
                self.partitioning.pop()
                self._emit(f"loop over {solution.text} {{\n  {solution.text} += ( {omega} * {correction.text} )\n}}")
                self._emit(f"apply bc to {solution.text}")
                self._emit(f"communicate {solution.text}")
                if correction.kind == "prolongated" and correction.level == level - 1:
                    self._op(abi.OP_PROLONG_ADD, level, abi.F_SOL, abi.F_SOL, omega=float(omega))
                else:
                    self._ops_ok = False
            return solution
        if kind == "Multiplication":
            operator, operand = expression.operand1, expression.operand2
            op_kind = type(operator).__name__
            if op_kind == "CoarseGridSolver":
                self._update_rhs(operand, level)
                self._emit(f"{self.fct_CGS}@{level} (  )")
                self._op(abi.OP_FAS_CGS, level, abi.F_SOL, 0, abi.F_RHS)
                return _field(self._names["solution"], level)
            if op_kind == "Inverse":
                self._smoothing(expression, level)
                return None
            if op_kind == "Operator":
                inner = self.traverse_graph(operand)
                self.traverse_graph(operator)
                text = self._operator_of(inner.text, level)
                if inner.kind == "field" and inner.name == self._names["approximation"]:
                    return _Term(text, "operator_of_approximation", level=level)
                return _Term(text)
            if op_kind == "Restriction" and ("Approximation" in type(operand).__name__ or type(operand).__name__ == "Cycle"):
                restriction = self.traverse_graph(operator)
                approximation, solution = self._n("approximation", level), self._n("solution", level)
                self._emit(f"loop over {approximation} {{\n  {approximation} = ( {restriction.text} * {solution} )\n"
                           f"  {solution} = {approximation}\n}}")
                self._emit(f"communicate {approximation}")
                self._emit(f"communicate {solution}")
                self._op(abi.OP_FAS_RESTRICT_SOLUTION, level, abi.F_CORR, abi.F_SOL)
                return _field(self._names["approximation"], level)
            a = self.traverse_graph(operator)
            b = self.traverse_graph(operand)
            text = f"( {a.text} * {b.text} )"
            if (a.kind == "stencil" and a.name == self._names["prolongation"] and b.kind == "field"
                    and b.name == self._names["solution"]):
                return _Term(text, "prolongated", level=b.level)
            if (a.kind == "stencil" and a.name == self._names["restriction"] and b.kind == "field"
                    and b.name == self._names["residual"]):
                return _Term(text, "restricted_residual", level=b.level)
            return _Term(text)
        if kind == "Addition":
            a = self.traverse_graph(expression.operand1)
            b = self.traverse_graph(expression.operand2)
            text = f"( {a.text} + {b.text} )"
            if a.kind == "restricted_residual" and b.kind == "operator_of_approximation" and a.level == b.level + 1:
                return _Term(text, "coarse_rhs", level=b.level)
            return _Term(text)
        if kind == "Subtraction":
            solution = self.traverse_graph(expression.operand1)
            approximation = self._n("approximation", level)
            self._emit(f"loop over {solution.text} {{\n  {solution.text} -= {approximation}\n}}")
            self._emit(f"communicate {solution.text}")
            self._op(abi.OP_FAS_SUBTRACT, level, abi.F_SOL, abi.F_CORR)
            return solution
        if "Residual" in kind:
            self._update_rhs(expression.rhs, level)
            self.traverse_graph(expression.operator)
            solution, residual = self._n("solution", level), self._n("residual", level)
            self._emit(f"loop over {residual} {{\n  {residual} = ( {self._n('rhs', level)} - {self._operator_of(solution, level)} )\n}}")
            self._emit(f"apply bc to {residual}")
            self._emit(f"communicate {residual}")
            self._op(abi.OP_FAS_RESIDUAL, level, abi.F_RES, abi.F_SOL, abi.F_RHS)
            return _field(self._names["residual"], level)
        if "Approximation" in kind:
            return _field(self._names["solution"], level)
        if kind == "RightHandSide":
            return _field(self._names["rhs"], level)
        if kind == "Prolongation":
            self.update_rhs[level - 1] = True           # the right-hand side below has to be formed again
            return _stencil(self._names["prolongation"], level - 1)
        if kind == "Restriction":
            return _stencil(self._names["restriction"], level + 1)
        if kind == "Operator":
            return _stencil(self._names["op_linear"], level)
        raise UnsupportedFASExpression(f"{kind} is not part of the FAS grammar's vocabulary")

    # ---- the reference's interface --------------------------------------------------------------------------------
    def _walk(self, expression):
        self._lines, self._ops, self._ops_ok = [], [], True
        self.traverse_graph(expression)

    def generate_mgfunction(self, expression):
        self._walk(expression)

    def generate_cycle_function(self, *args):
        expression = next(a for a in args if type(a).__name__ == "Cycle")
        self._walk(expression)
        return "Function " + self.fct_name + " (  ) {\n" + "\n".join(self._lines) + "\n}\n\n"

    def generate_op_program(self, expression):
        """The flat op program of the cycle (EVOMG_OP_FAS_*); raises when the DAG uses a shape only the text form covers."""
        self._walk(expression)
        if not self._ops_ok:
            raise UnsupportedFASExpression("the cycle contains a statement outside the FAS op vocabulary")
        return list(self._ops)

    def generate_storage(self, *args):
        return []

    def initialize_code_generation(self, *args):
        if self._handle is None and self._library is not None:
            pd = self.problem_description()
            self._handle = abi.Handle(self._library, pd.to_c(), self._device_id or 0)
            pd.upload_initial_fields(self._handle)
        return 0.0, 0.0

    def reinitialize(self, *args):
        return None

    def problem_description(self):
        from evostencils_amd import problems
        return problems.fas_2d(self.template, self.problem_name)

    @staticmethod
    def _printed(value):
        """The residual as `Solve` prints it: std::cout precision 4 / 3 / 2 / 1 by magnitude, nothing at all at or below
        1e-12 (template:118-160)."""
        if value <= 1.0e-12:
            return None
        digits = 1 if value <= 1.0e-11 else 2 if value <= 1.0e-10 else 3 if value <= 1.0e-9 else 4
        return float(f"%.{digits}g" % value)

    def interpret_history(self, history, time_ms, infinity=1e100):
        """(t, c, n) as exastencils_FAS.py:370-394 derives them from the executable's stdout."""
        res_initial = self._printed(history[0]) or 0.0
        res_final = 0.0
        for value in history[1:]:
            printed = self._printed(value)
            if printed is not None:
                res_final = printed
        n = len(history) - 1
        try:
            c = (res_final / res_initial) ** (1.0 / n)
        except (ZeroDivisionError, OverflowError):
            return infinity, infinity, infinity
        if math.isinf(c) or math.isnan(c):
            return infinity, infinity, infinity
        return time_ms, c, n

    def generate_and_evaluate(self, *args, **kwargs):
        expression = next((a for a in args if type(a).__name__ == "Cycle"), None)
        evaluation_samples = kwargs.get("evaluation_samples", 1)
        infinity = 1e100
        try:
            ops = self.generate_op_program(expression)
        except UnsupportedFASExpression:
            return infinity, infinity, infinity
        self.initialize_code_generation()
        if self._handle is None:
            raise RuntimeError("ProgramGeneratorFAS: no HIP library (there is no CPU fallback)")
        ts, cs, ns = [], [], []
        for _ in range(evaluation_samples):
            status, out = self._handle.try_evaluate(ops, 1, 0.0, self.template.max_iterations + 2)
            if status == abi.ERR_DEVICE:
                raise abi.BackendError(self._handle.last_error())
            if status != abi.OK or out is None:
                return infinity, infinity, infinity
            t, c, n = self.interpret_history(list(out.history), out.time_ms, infinity)
            ts.append(t), cs.append(c), ns.append(n)
        return mean(ts), mean(cs), mean(ns)

    def close(self):
        if self._handle is not None:
            self._handle.close()
            self._handle = None
