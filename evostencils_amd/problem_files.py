"""Problem-file front end: reads the reference's problem directories (`.settings`, `.knowledge`, layer-2 `.exa2`,
layer-3 solver hints `.exa3`) DIRECTLY into a `problems.ProblemDescription`.

The reference cannot do this itself: it runs the ExaStencils compiler once and parses the *debug* layer-3 file that
compiler writes (code_generation/parser.py:25-111 `extract_l2_information`, :114-125 `extract_knowledge_information`,
:128-143 `extract_settings_information`). Here the JVM step is gone, so the information is taken from the sources the
user wrote. Supported: one scalar `Real` field on the nodes of one rectangular domain, one constant-coefficient star
stencil written in terms of `vf_gridWidth_*` (rediscretised on every level, like ExaStencils does for such stencils),
one equation `Op * u == RHS`; Dirichlet boundary values and right-hand side given as arithmetic expressions of the
node position; systems of two such fields (`describe_system`); and — round 3 — complex Helmholtz-type problems that enter at
layer 3 (`describe_layer3_helmholtz`: the `.exa3` / `.exa4` pair of the reference's Helmholtz example and problems of the same
shape). Everything else raises `UnsupportedProblem` with the construct that was not understood.

Layer-2 statements (keywords `Domain`, `Field`, `Operator`, `Equation` are optional — the reference's own
Poisson/2D_FD_Poisson_fromL2.exa2 omits them):

    [Domain] <name>< [a, b(, c)] to [d, e(, f)] >
    [Field] <f> with Real on Node of <domain> = <expr>
    [Field] <f>[@<levels>] on boundary = <expr>          <levels>: finest | coarsest | all | (all but finest) | <int>
    [Operator] <Op> from Stencil { [o0, o1(, o2)] => <expr> ... }
    [Equation] <eq> { <Op> * <f> == <rhs field> }

Expressions: numbers, `PI`, `vf_nodePos_{x,y,z}`, `vf_boundaryPos_*` / `vf_boundaryCoord_*` (the node's own position
for node-based fields), `vf_gridWidth_*`, `+ - * / **`, unary minus, parentheses, sin cos tan exp sqrt log abs.
They are evaluated by a small recursive-descent evaluator with Python's precedence and associativity (no `eval`),
one numpy operation per operator, so a value computed from the file is bit-identical to the same formula written in
Python.
"""
import os
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from evostencils_amd import abi, problems


class UnsupportedProblem(RuntimeError):
    pass


# ---- key = value files (.settings / .knowledge) ----------------------------------------------------------------------
def _strip_comment(line):
    i = line.find("//")
    return line if i < 0 else line[:i]


def read_key_values(path, _seen=None) -> Dict[str, str]:
    """`key = value` lines; `import '<relative path>'` pulls in another file first (knowledge files do that, e.g.
    Poisson/2D_FD_Poisson_fromL2.knowledge:9-10); `//` comments; later assignments win."""
    path = os.path.abspath(path)
    seen = _seen if _seen is not None else set()
    if path in seen:
        raise UnsupportedProblem(f"circular import of {path}")
    seen.add(path)
    out: Dict[str, str] = {}
    with open(path, "r") as fh:
        for raw in fh:
            line = _strip_comment(raw).strip()
            if not line:
                continue
            m = re.match(r"import\s+['\"](.+?)['\"]", line)
            if m:
                out.update(read_key_values(os.path.join(os.path.dirname(path), m.group(1)), seen))
                continue
            if "=" in line:
                k, v = line.split("=", 1)
                out[k.strip()] = v.strip().strip('"')
    return out


def resolve_settings(settings: Dict[str, str]) -> Dict[str, str]:
    """`$key$` placeholders (e.g. l2file = "$configName$.exa2", settings:7-9)."""
    out = dict(settings)
    for _ in range(4):
        for k, v in out.items():
            out[k] = re.sub(r"\$(\w+)\$", lambda m: out.get(m.group(1), m.group(0)), v)
    return out


# ---- expressions -----------------------------------------------------------------------------------------------------
_TOKEN = re.compile(r"\s*(?:((?:\d+\.\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|\d+(?:[eE][+-]?\d+)?)j?)|([A-Za-z_]\w*)|(\*\*|[-+*/(),]))")
_FUNCTIONS = {"sin": np.sin, "cos": np.cos, "tan": np.tan, "exp": np.exp, "sqrt": np.sqrt, "log": np.log,
              "abs": np.abs, "fabs": np.abs}
_FUNCTIONS2 = {"max": np.maximum, "min": np.minimum}        # two arguments (Helmholtz right-hand side, exa3:24)


def tokenize(text: str) -> List[Tuple[str, str]]:
    pos, out = 0, []
    text = text.strip()
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise UnsupportedProblem(f"cannot read expression at: {text[pos:pos + 24]!r}")
        if m.group(1) is not None:
            out.append(("num", m.group(1)))
        elif m.group(2) is not None:
            out.append(("id", m.group(2)))
        else:
            out.append(("op", m.group(3)))
        pos = m.end()
    return out


class Expression:
    """Parsed arithmetic expression. Grammar (Python's): sum := term (('+'|'-') term)* ; term := factor (('*'|'/')
    factor)* ; factor := ('+'|'-') factor | power ; power := atom ('**' factor)? ; atom := num | id | id '(' sum ')'
    | '(' sum ')'."""

    def __init__(self, text: str):
        self.text = text.strip()
        self._tokens = tokenize(self.text)
        self._pos = 0
        self.tree = self._sum()
        if self._pos != len(self._tokens):
            raise UnsupportedProblem(f"trailing input in expression {self.text!r}")

    # -- parser
    def _peek(self):
        return self._tokens[self._pos] if self._pos < len(self._tokens) else (None, None)

    def _take(self, kind=None, value=None):
        k, v = self._peek()
        if k is None or (kind and k != kind) or (value and v != value):
            raise UnsupportedProblem(f"unexpected {v!r} in expression {self.text!r}")
        self._pos += 1
        return v

    def _sum(self):
        node = self._term()
        while self._peek() in (("op", "+"), ("op", "-")):
            op = self._take()
            node = (op, node, self._term())
        return node

    def _term(self):
        node = self._factor()
        while self._peek() in (("op", "*"), ("op", "/")):
            op = self._take()
            node = (op, node, self._factor())
        return node

    def _factor(self):
        if self._peek() in (("op", "+"), ("op", "-")):
            op = self._take()
            return ("neg" if op == "-" else "pos", self._factor())
        return self._power()

    def _power(self):
        base = self._atom()
        if self._peek() == ("op", "**"):
            self._take()
            return ("**", base, self._factor())
        return base

    def _atom(self):
        k, v = self._peek()
        if k == "num":
            self._take()
            return ("num", complex(0.0, float(v[:-1])) if v.endswith("j") else float(v))     # 0.5j: imaginary literal
        if k == "id":
            self._take()
            if self._peek() == ("op", "("):
                if v not in _FUNCTIONS and v not in _FUNCTIONS2:
                    raise UnsupportedProblem(f"unknown function {v!r} in expression {self.text!r}")
                self._take()
                arg = self._sum()
                if v in _FUNCTIONS2:
                    self._take("op", ",")
                    arg2 = self._sum()
                    self._take("op", ")")
                    return ("call2", v, arg, arg2)
                self._take("op", ")")
                return ("call", v, arg)
            return ("id", v)
        if (k, v) == ("op", "("):
            self._take()
            node = self._sum()
            self._take("op", ")")
            return node
        raise UnsupportedProblem(f"unexpected {v!r} in expression {self.text!r}")

    # -- evaluation
    def names(self):
        found = set()

        def walk(n):
            if n[0] == "id":
                found.add(n[1])
            elif n[0] in ("neg", "pos"):
                walk(n[1])
            elif n[0] == "call":
                walk(n[2])
            elif n[0] == "call2":
                walk(n[2])
                walk(n[3])
            elif n[0] != "num":
                walk(n[1])
                walk(n[2])
        walk(self.tree)
        return found

    def evaluate(self, env: Dict[str, object]):
        def ev(n):
            t = n[0]
            if t == "num":
                return n[1]
            if t == "id":
                if n[1] not in env:
                    raise UnsupportedProblem(f"unknown symbol {n[1]!r} in expression {self.text!r}")
                return env[n[1]]
            if t == "neg":
                return -ev(n[1])
            if t == "pos":
                return ev(n[1])
            if t == "call":
                return _FUNCTIONS[n[1]](ev(n[2]))
            if t == "call2":
                return _FUNCTIONS2[n[1]](ev(n[2]), ev(n[3]))
            a, b = ev(n[1]), ev(n[2])
            if t == "+":
                return a + b
            if t == "-":
                return a - b
            if t == "*":
                return a * b
            if t == "/":
                return a / b
            return a ** b
        return ev(self.tree)


# ---- layer 2 ---------------------------------------------------------------------------------------------------------
@dataclass
class Layer2:
    domain_name: str = ""
    lower: Tuple[float, ...] = ()
    upper: Tuple[float, ...] = ()
    fields: Dict[str, Tuple[str, str, Expression]] = field(default_factory=dict)       # name -> (type, where, init)
    boundaries: Dict[str, List[Tuple[str, Expression]]] = field(default_factory=dict)  # name -> [(levels, expr)]
    operators: Dict[str, List[Tuple[Tuple[int, ...], Expression]]] = field(default_factory=dict)
    equations: Dict[str, Tuple[str, str, str]] = field(default_factory=dict)           # name -> (op, field, rhs)
    # systems of PDEs: name -> (left-hand side text, rhs field) of equations that are not `Op * field == rhs`
    general_equations: Dict[str, Tuple[str, str]] = field(default_factory=dict)
    globals_: Dict[str, Expression] = field(default_factory=dict)                      # `Globals { Expr name = value }`


_NUM = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?"


def parse_layer2(text: str) -> Layer2:
    out = Layer2()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)             # block comments (LinearElasticity/...exa2:38-43)
    src = "\n".join(_strip_comment(l) for l in text.splitlines())
    # blocks first: Operator ... from Stencil { ... } and Equation ... { ... }
    def take_blocks(pattern, handler, s):
        while True:
            m = re.search(pattern, s)
            if not m:
                return s
            depth, i = 1, m.end()
            while i < len(s) and depth:
                depth += {"{": 1, "}": -1}.get(s[i], 0)
                i += 1
            if depth:
                raise UnsupportedProblem("unbalanced braces in layer-2 file")
            handler(m, s[m.end():i - 1])
            s = s[:m.start()] + "\n" + s[i:]

    def on_stencil(m, body):
        entries = []
        for line in body.splitlines():
            line = line.strip()
            if not line:
                continue
            e = re.match(r"\[([^\]]*)\]\s*=>\s*(.+)$", line)
            if not e:
                raise UnsupportedProblem(f"stencil entry not understood: {line!r}")
            entries.append((tuple(int(c) for c in e.group(1).split(",")), Expression(e.group(2))))
        out.operators[m.group(1)] = entries

    def on_equation(m, body):
        if m.group(1) == "Globals":
            for line in body.splitlines():
                line = line.strip()
                if not line:
                    continue
                g = re.match(r"(?:Expr|Val|Var)\s+(\w+)\s*(?::\s*\w+\s*)?=\s*(.+)$", line)
                if not g:
                    raise UnsupportedProblem(f"Globals entry not understood: {line!r}")
                out.globals_[g.group(1)] = Expression(g.group(2))
            return
        if m.group(1) in ("Function", "Solve"):
            raise UnsupportedProblem(f"layer-2 `{m.group(1)} {{...}}` block is not understood")
        flat = " ".join(body.split())
        e = re.match(r"\s*(\w+)\s*\*\s*(\w+)\s*==\s*(\w+)\s*$", flat)
        if e:
            out.equations[m.group(1)] = (e.group(1), e.group(2), e.group(3))
            return
        g = re.match(r"(.+)==\s*(\w+)\s*$", flat)
        if not g:
            raise UnsupportedProblem(f"equation {m.group(1)!r}: `<linear expression> == rhs_field` expected, got {flat!r}")
        out.general_equations[m.group(1)] = (g.group(1).strip(), g.group(2))

    src = take_blocks(r"(?:\bOperator\s+)?\b(\w+)\s+from\s+Stencil\s*\{", on_stencil, src)
    src = take_blocks(r"(?:\bEquation\s+)?\b(\w+)\s*\{", on_equation, src)
    for raw in src.splitlines():
        line = raw.strip()
        if not line:
            continue
        m = re.match(rf"(?:Domain\s+)?(\w+)\s*<\s*\[([^\]]*)\]\s*to\s*\[([^\]]*)\]\s*>\s*$", line)
        if m:
            out.domain_name = m.group(1)
            out.lower = tuple(float(c) for c in m.group(2).split(","))
            out.upper = tuple(float(c) for c in m.group(3).split(","))
            continue
        m = re.match(r"(?:Field\s+)?(\w+)\s+with\s+(\w+(?:\s*<\s*\w+\s*>)?)\s+on\s+(\w+)\s+of\s+(\w+)\s*(?:=\s*(.+))?$", line)
        if m:
            out.fields[m.group(1)] = (m.group(2).replace(" ", ""), m.group(3), Expression(m.group(5) or "0.0"))
            continue
        m = re.match(r"(?:Field\s+)?(\w+)\s*(?:@\s*(\([^)]*\)|\w+))?\s+on\s+boundary\s*=\s*(.+)$", line)
        if m:
            levels = " ".join((m.group(2) or "all").strip("()").split())
            out.boundaries.setdefault(m.group(1), []).append((levels, Expression(m.group(3))))
            continue
        raise UnsupportedProblem(f"layer-2 statement not understood: {line!r}")
    return out


# ---- layer-3 solver hints ------------------------------------------------------------------------------------------
def parse_solver_hints(text: str) -> Dict[str, str]:
    """`generate solver for u in solEq with { key = value ... }` (Poisson/2D_FD_Poisson_fromL2.exa3:1-15)."""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    src = "\n".join(_strip_comment(l) for l in text.splitlines())
    m = re.search(r"generate\s+solver\s+for\s+((?:\w+\s+in\s+\w+)(?:\s+and\s+\w+\s+in\s+\w+)*)\s+with\s*\{([^}]*)\}", src)
    if not m:
        return {}
    pairs = re.findall(r"(\w+)\s+in\s+(\w+)", m.group(1))
    hints = {"field": pairs[0][0], "equation": pairs[0][1]}
    if len(pairs) > 1:          # a system of PDEs: unknown i belongs to equation i (LinearElasticity/...exa3:1)
        hints["fields"] = ",".join(f for f, _ in pairs)
        hints["equations"] = ",".join(e for _, e in pairs)
    for line in m.group(2).splitlines():
        if "=" in line:
            k, v = line.split("=", 1)
            hints[k.strip()] = v.strip().strip('"')
    return hints


# ---- assembling the descriptor ---------------------------------------------------------------------------------------
_AXES = "xyz"


def _star_slot(offset):
    nz = [(d, c) for d, c in enumerate(offset) if c != 0]
    if not nz:
        return 0
    if len(nz) == 1 and nz[0][1] in (-1, 1):
        d, c = nz[0]
        return 1 + 2 * d + (1 if c > 0 else 0)
    return None


def _applies(levels: str, level: int, min_level: int, max_level: int) -> bool:
    if levels == "all":
        return True
    if levels == "finest":
        return level == max_level
    if levels == "coarsest":
        return level == min_level
    m = re.match(r"all but (\w+)$", levels)
    if m:
        return not _applies(m.group(1), level, min_level, max_level)
    if levels.isdigit():
        return level == int(levels)
    raise UnsupportedProblem(f"level specification {levels!r} not understood")


def describe(l2: Layer2, name: str, dimension: int, min_level: int, max_level: int,
             hints: Optional[Dict[str, str]] = None) -> problems.ProblemDescription:
    hints = hints or {}
    if len(l2.lower) != dimension or len(l2.upper) != dimension:
        raise UnsupportedProblem(f"domain has {len(l2.lower)} dimensions, knowledge says {dimension}")
    if dimension not in (2, 3):
        raise UnsupportedProblem("only 2D and 3D problems are supported")
    if l2.general_equations or len(l2.equations) > 1:
        return describe_system(l2, name, dimension, min_level, max_level, hints)
    if len(l2.equations) != 1:
        raise UnsupportedProblem(f"no equation found in the layer-2 file")
    (eq_name, (op_name, field_name, rhs_name)), = l2.equations.items()
    if hints.get("equation", eq_name) != eq_name or hints.get("field", field_name) != field_name:
        raise UnsupportedProblem("the generated solver targets another field/equation than the layer-2 file defines")
    for f in (field_name, rhs_name):
        if f not in l2.fields:
            raise UnsupportedProblem(f"field {f!r} is not declared")
        kind, where, _ = l2.fields[f]
        if kind != "Real" or where != "Node":
            raise UnsupportedProblem(f"field {f!r}: only `Real on Node` is supported, got `{kind} on {where}`")
    if op_name not in l2.operators:
        raise UnsupportedProblem(f"operator {op_name!r} is not declared")
    # A stencil that is not a star (code_generation/parser.py:8-22 takes any offsets): in 2D, inside the 3 x 3 neighbourhood, the
    # problem runs as a "system" of ONE unknown — the path of the coupled problems carries compact nine-point stencils
    # (csrc/impl_system.hip); the tuned strip kernels of the scalar path are written for stars.
    if dimension == 2 and any(len(off) == dimension and _star_slot(off) is None and all(abs(c) <= 1 for c in off)
                              for off, _ in l2.operators[op_name]):
        return describe_system(l2, name, dimension, min_level, max_level, hints)
    entries = {}
    for offset, expr in l2.operators[op_name]:
        slot = _star_slot(offset) if len(offset) == dimension else None
        if slot is None:
            raise UnsupportedProblem(f"operator {op_name!r}: offset {list(offset)} is outside the star stencil")
        if slot in entries:
            raise UnsupportedProblem(f"operator {op_name!r}: offset {list(offset)} given twice")
        allowed = {"PI"} | {f"vf_gridWidth_{a}" for a in _AXES[:dimension]}
        if not expr.names() <= allowed:
            raise UnsupportedProblem(f"operator {op_name!r}: only constant coefficients in vf_gridWidth_* are "
                                     f"supported, found {sorted(expr.names() - allowed)}")
        entries[slot] = expr
    if 0 not in entries:
        raise UnsupportedProblem(f"operator {op_name!r} has no centre entry")
    lower, upper = l2.lower, l2.upper

    def widths(level):
        return [(upper[d] - lower[d]) / (1 << level) for d in range(dimension)]

    def stencil(level, _params):
        env = {"PI": np.pi}
        for d, h in enumerate(widths(level)):
            env[f"vf_gridWidth_{_AXES[d]}"] = h
        return [float(entries[s].evaluate(env)) if s in entries else 0.0 for s in range(7)]

    def position_env(level):
        n = (1 << level) + 1
        env = {"PI": np.pi}
        for d, h in enumerate(widths(level)):
            shape = [1] * dimension
            shape[dimension - 1 - d] = n                      # arrays are [i2][i1][i0], x fastest
            x = (lower[d] + np.arange(n, dtype=np.float64) * h).reshape(shape)
            for prefix in ("vf_nodePos_", "vf_boundaryPos_", "vf_boundaryCoord_", "vf_nodePosition_"):
                env[prefix + _AXES[d]] = x
            env[f"vf_gridWidth_{_AXES[d]}"] = h
        return n, env

    def on_grid(expr, level):
        n, env = position_env(level)
        return np.ascontiguousarray(np.broadcast_to(np.asarray(expr.evaluate(env), dtype=np.float64), (n,) * dimension))

    def boundary_expr(level):
        # the descriptor's initial solution is only ever asked for the finest level of a run (coarser levels hold
        # error-type fields with homogeneous boundary values), and that level moves when the optimizer bumps the
        # problem size (reinitialize), so "finest" means the level asked for
        chosen = None
        for levels, expr in l2.boundaries.get(field_name, []):
            if _applies(levels, level, min(min_level, level), level):
                chosen = expr
        return chosen

    def initial_solution(level):
        n = (1 << level) + 1
        u = on_grid(l2.fields[field_name][2], level).copy()
        bc = boundary_expr(level)
        if bc is not None:
            mask = problems._boundary_mask(n, dimension)
            u[mask] = on_grid(bc, level)[mask]
        return u

    def right_hand_side(level):
        return on_grid(l2.fields[rhs_name][2], level)

    cgs = hints.get("solver_cgs", "CG")
    if cgs not in ("CG", "BiCGStab", "BiCGSTAB"):
        raise UnsupportedProblem(f"coarse-grid solver {cgs!r} is not supported")
    return problems.ProblemDescription(
        name, dimension, min_level, max_level,
        cgs_kind=abi.CGS_CG if cgs == "CG" else abi.CGS_BICGSTAB,
        max_iterations=int(hints.get("solver_maxNumIts", 100)),
        cgs_max_iterations=int(hints.get("solver_cgs_maxNumIts", 1000)),
        target_res_reduction=float(hints.get("solver_targetResReduction", 1e-12)),
        cgs_target_res_reduction=float(hints.get("solver_cgs_targetResReduction", 1e-12)),
        field_name=field_name, rhs_name=rhs_name, operator_name=op_name, equation_name=eq_name,
        restriction_name=f"gen_restrictionForRes_{field_name}", prolongation_name=f"gen_prolongationForSol_{field_name}",
        mg_stencil=stencil, initial_solution=initial_solution, right_hand_side=right_hand_side,
        domain_lower=tuple(lower) + (0.0,) * (3 - dimension), domain_upper=tuple(upper) + (1.0,) * (3 - dimension),
        solver_hints={k: v for k, v in hints.items() if k.startswith("solver_")})


def describe_system(l2: Layer2, name: str, dimension: int, min_level: int, max_level: int,
                    hints: Optional[Dict[str, str]] = None) -> problems.ProblemDescription:
    """Systems of PDEs (LinearElasticity/2D_FD_LinearElasticity_fromL2.exa2): several `Real on Node` unknowns,
    equations whose left-hand sides are linear combinations `coefficient * Stencil * field` with coefficients built from
    numbers and `Globals`, named stencils with constant coefficients in vf_gridWidth_* (any offsets of the 3 x 3
    neighbourhood). Which unknown belongs to which equation comes from the layer-3 `generate solver for u in uEq and v in
    vEq` line, else from the order of declaration."""
    hints = hints or {}
    if dimension != 2:
        raise UnsupportedProblem("systems of PDEs are supported in 2D only")
    equations = dict(l2.general_equations)
    for eq_name, (op, fld, rhs) in l2.equations.items():
        equations[eq_name] = (f"{op} * {fld}", rhs)
    if "fields" in hints:
        field_names, equation_names = hints["fields"].split(","), hints["equations"].split(",")
    else:
        equation_names = list(equations)
        rhs_fields = {rhs for _, rhs in equations.values()}
        field_names = [f for f in l2.fields if f not in rhs_fields][:len(equation_names)]
    if sorted(equation_names) != sorted(equations) or len(field_names) != len(equation_names):
        raise UnsupportedProblem(f"equations {sorted(equations)} do not match the solver's {equation_names}")
    if len(field_names) > abi.MAX_UNKNOWNS:
        raise UnsupportedProblem(f"at most {abi.MAX_UNKNOWNS} unknowns per node are supported, found {field_names}")
    rhs_names = [equations[e][1] for e in equation_names]
    for f in field_names + rhs_names:
        if f not in l2.fields:
            raise UnsupportedProblem(f"field {f!r} is not declared")
        kind, where, _ = l2.fields[f]
        if kind != "Real" or where != "Node":
            raise UnsupportedProblem(f"field {f!r}: only `Real on Node` is supported, got `{kind} on {where}`")
    lower, upper = l2.lower, l2.upper
    allowed = {"PI"} | {f"vf_gridWidth_{a}" for a in _AXES[:dimension]}
    for op_name, entries in l2.operators.items():
        for offset, expr in entries:
            if len(offset) != dimension or any(abs(c) > 1 for c in offset):
                raise UnsupportedProblem(f"operator {op_name!r}: offset {list(offset)} is outside the 3 x 3 neighbourhood")
            if not expr.names() <= allowed:
                raise UnsupportedProblem(f"operator {op_name!r}: only constant coefficients in vf_gridWidth_* are supported, "
                                         f"found {sorted(expr.names() - allowed)}")
    global_values = {}
    for g, expr in l2.globals_.items():
        if not expr.names() <= {"PI"}:
            raise UnsupportedProblem(f"global {g!r}: only numeric expressions are supported")
        global_values[g] = float(expr.evaluate({"PI": np.pi}))

    def widths(level):
        return [(upper[d] - lower[d]) / (1 << level) for d in range(dimension)]

    def operator_stencils(level):
        env = {"PI": np.pi}
        for d, h in enumerate(widths(level)):
            env[f"vf_gridWidth_{_AXES[d]}"] = h
        return {op: [(tuple(off), float(e.evaluate(env))) for off, e in entries] for op, entries in l2.operators.items()}

    def equation_text(level):
        """The equation with globals numeric and level tags on stencils and fields, every token blank-separated (what
        grammar.multigrid.EquationInfo parses; the reference gets this text from ExaStencils' debug layer 3)."""
        texts = []
        for eq in equation_names:
            lhs, rhs = equations[eq]

            def token(m):
                w = m.group(0)
                if w in l2.operators or w in l2.fields:
                    return f"{w}@{level}"
                if w in global_values:
                    return repr(global_values[w])
                raise UnsupportedProblem(f"equation {eq!r}: unknown symbol {w!r}")
            spaced = re.sub(r"([()*+\-/])", r" \1 ", lhs)
            body = " ".join(re.sub(r"[A-Za-z_]\w*", token, spaced).split())
            texts.append(f"( {body} ) == {rhs}@{level}")
        return texts

    def position_env(level):
        n = (1 << level) + 1
        env = {"PI": np.pi}
        for d, h in enumerate(widths(level)):
            shape = [1] * dimension
            shape[dimension - 1 - d] = n
            x = (lower[d] + np.arange(n, dtype=np.float64) * h).reshape(shape)
            for prefix in ("vf_nodePos_", "vf_boundaryPos_", "vf_boundaryCoord_", "vf_nodePosition_"):
                env[prefix + _AXES[d]] = x
            env[f"vf_gridWidth_{_AXES[d]}"] = h
        env.update(global_values)
        return n, env

    def on_grid(expr, level):
        n, env = position_env(level)
        return np.ascontiguousarray(np.broadcast_to(np.asarray(expr.evaluate(env), dtype=np.float64), (n,) * dimension))

    def initial_solution(level):
        n = (1 << level) + 1
        out = []
        for f in field_names:
            u = on_grid(l2.fields[f][2], level).copy()
            chosen = None
            for levels, expr in l2.boundaries.get(f, []):
                if _applies(levels, level, min(min_level, level), level):
                    chosen = expr
            if chosen is not None:
                mask = problems._boundary_mask(n, dimension)
                u[mask] = on_grid(chosen, level)[mask]
            out.append(u)
        return out

    def right_hand_side(level):
        return [on_grid(l2.fields[r][2], level) for r in rhs_names]

    cgs = hints.get("solver_cgs", "CG")
    if cgs != "CG":
        raise UnsupportedProblem(f"systems of PDEs: coarse-grid solver {cgs!r} is not supported")
    spec = problems.SystemSpec(list(field_names), rhs_names, list(equation_names), equation_text, operator_stencils,
                               [f"gen_restrictionForRes_{f}" for f in field_names],
                               [f"gen_prolongationForSol_{f}" for f in field_names], initial_solution, right_hand_side)
    try:
        spec.assembled(max_level)      # every term must be a scaled named stencil applied to an unknown
    except (ValueError, KeyError, StopIteration) as exc:
        raise UnsupportedProblem(f"system of PDEs: {exc}") from exc
    return problems.ProblemDescription(
        name, dimension, min_level, max_level, system=spec, parameters=dict(global_values),
        max_iterations=int(hints.get("solver_maxNumIts", 100)),
        cgs_max_iterations=int(hints.get("solver_cgs_maxNumIts", 1000)),
        target_res_reduction=float(hints.get("solver_targetResReduction", 1e-12)),
        cgs_target_res_reduction=float(hints.get("solver_cgs_targetResReduction", 1e-12)),
        domain_lower=tuple(lower) + (0.0,) * (3 - dimension), domain_upper=tuple(upper) + (1.0,) * (3 - dimension),
        solver_hints={k: v for k, v in hints.items() if k.startswith("solver_")})


# ---- problems that enter at layer 3: complex Helmholtz-type problems with a hand-written Krylov solver ------------------
def _function_body(src: str, name: str) -> Optional[str]:
    """Text between the braces of `Function <name> ... {`, None if there is no such function."""
    m = re.search(r"\bFunction\s+" + re.escape(name) + r"\b[^{]*\{", src)
    if not m:
        return None
    depth, i = 1, m.end()
    while i < len(src) and depth:
        depth += {"{": 1, "}": -1}.get(src[i], 0)
        i += 1
    return src[m.end():i - 1]


def _loop_limits(body: str, what: str) -> Tuple[int, float]:
    """(`repeat N times count ...`, the factor in `fabs(curRes) < F * fabs(initRes)`) of a hand-written solver loop."""
    r = re.search(r"\brepeat\s+(\d+)\s+times\s+count\b", body)
    t = re.search(r"fabs\s*\(\s*curRes\s*\)\s*<\s*(" + _NUM + r")\s*\*\s*fabs\s*\(\s*initRes\s*\)", body)
    if not r or not t:
        raise UnsupportedProblem(f"{what}: `repeat N times count ...` with `fabs(curRes) < F * fabs(initRes)` expected")
    return int(r.group(1)), float(t.group(1))


def describe_layer3_helmholtz(l3_text: str, l4_text: str, name: str, dimension: int, min_level: int, max_level: int
                              ) -> problems.ProblemDescription:
    """Layer-3 entry of the reference's Helmholtz example (Helmholtz/2D_FD_Helmholtz_fromL3.exa3/.exa4) and problems of
    the same shape: ONE complex node field; the outer operator and the operator the multigrid cycle acts on as star
    stencils over `vf_gridWidth_*` and `Globals` (k, shift, ...); `Equation <eq> { <M> * <u> == <f> }`; the right-hand
    side `Field RHS@finest ... = <expr>`; boundaries through `applyBC_<u>` functions on layer 4 that are the first-order
    Robin condition left / right and homogeneous Dirichlet top / bottom (exa4:43-59); `Solve@finest` = the hand-written
    preconditioned BiCGStab (exa3:144-200) and `gen_mgCycle@coarsest` = the hand-written BiCGStab (exa3:396-433) — their
    texts ARE the algorithms the backend implements, so only their iteration caps and tolerances are read, and everything
    else about them is checked for presence, not interpreted. Anything that does not fit raises UnsupportedProblem."""
    if dimension != 2:
        raise UnsupportedProblem("layer-3 Helmholtz-type problems are 2D")
    text = re.sub(r"/\*.*?\*/", "", l3_text, flags=re.S)
    src = "\n".join(_strip_comment(l) for l in text.splitlines())
    l4 = "\n".join(_strip_comment(l) for l in re.sub(r"/\*.*?\*/", "", l4_text, flags=re.S).splitlines())

    def stencil(op_name):
        m = re.search(r"\bOperator\s+" + re.escape(op_name) + r"\s+from\s+Stencil\s*\{([^}]*)\}", src)
        if not m:
            raise UnsupportedProblem(f"operator {op_name!r} is not a `from Stencil` operator")
        entries = {}
        for line in m.group(1).splitlines():
            line = line.strip()
            if not line:
                continue
            e = re.match(r"\[([^\]]*)\]\s*=>\s*(.+)$", line)
            if not e:
                raise UnsupportedProblem(f"stencil entry not understood: {line!r}")
            off = tuple(int(c) for c in e.group(1).split(","))
            if _star_slot(off) is None or off in entries:
                raise UnsupportedProblem(f"operator {op_name!r}: only 5-point star stencils are supported")
            entries[off] = Expression(e.group(2))
        return entries

    eq = re.search(r"\bEquation\s+(\w+)\s*\{\s*(\w+)\s*\*\s*(\w+)\s*==\s*(\w+)\s*\}", src)
    if not eq:
        raise UnsupportedProblem("`Equation <name> { <Op> * <field> == <rhs> }` expected")
    eq_name, mg_op, field_name, rhs_name = eq.groups()
    fld = re.search(r"\bField\s+" + re.escape(field_name) + r"\s+with\s+Complex\s*<\s*Real\s*>\s+on\s+Node\b", src)
    if not fld:
        raise UnsupportedProblem(f"field {field_name!r}: `with Complex<Real> on Node` expected")
    glob = re.search(r"\bGlobals\s*\{([^}]*)\}", src)
    globals_ = {}
    if glob:
        for line in glob.group(1).splitlines():
            g = re.match(r"\s*(?:Expr|Val|Var)\s+(\w+)\s*(?::\s*[\w<>]+\s*)?=\s*(.+)$", line)
            if g:
                globals_[g.group(1)] = Expression(g.group(2))
    values = {}
    for gname, expr in globals_.items():               # in file order: later globals may use earlier ones
        values[gname] = expr.evaluate(dict(values, PI=np.pi))
    if "k" not in values:
        raise UnsupportedProblem("a Helmholtz-type problem needs the wavenumber `Expr k = ...` among its Globals")
    # outer solver
    solve = _function_body(src, "Solve@finest")
    call = re.match(r"\s*(\w+)\s*@\s*finest\s*\(\s*\)\s*$", " ".join((solve or "").split()))
    if not call:
        raise UnsupportedProblem("`Function Solve@finest { <solver>@finest ( ) }` expected")
    outer = _function_body(src, call.group(1) + "@finest")
    if outer is None or "gen_mgCycle" not in outer and "VCycle" not in outer and "mgCycle" not in outer:
        raise UnsupportedProblem(f"outer solver {call.group(1)!r}: a multigrid-preconditioned Krylov loop expected")
    m_outer = re.search(r"=\s*RHS\s*-\s*(\w+)\s*\*\s*Solution", outer)
    if not m_outer or "resHat" not in outer:
        raise UnsupportedProblem("outer solver: `Residual = RHS - <A> * Solution` of the preconditioned BiCGStab expected")
    outer_op = m_outer.group(1)
    max_its, tol = _loop_limits(outer, "outer solver")
    coarse = _function_body(src, "gen_mgCycle@coarsest")
    if coarse is None or "resHat" not in coarse:
        raise UnsupportedProblem("`Function gen_mgCycle@coarsest` = the hand-written BiCGStab expected")
    cgs_its, cgs_tol = _loop_limits(coarse, "coarse-grid solver")
    # boundary: the first-order Robin condition x_b = x_in / (1 - i k h) left / right, zero top / bottom
    bc = _function_body(l4, f"applyBC_{field_name}@all")
    robin = re.compile(r"\[\s*0\s*,\s*0\s*\]\s*=\s*\w+@\[\s*-?1\s*,\s*0\s*\]\s*/\s*\(\s*1\.0\s*-\s*\(\s*0\.0\s*\+\s*1\.0j\s*\)\s*\*\s*k\s*\*\s*vf_gridWidth_x\s*\)")
    if bc is None or len(robin.findall(bc)) != 2 or len(re.findall(r"\[\s*0\s*,\s*0\s*\]\s*=\s*0\.0", bc)) != 2:
        raise UnsupportedProblem(f"applyBC_{field_name}: first-order Robin left / right and zero top / bottom expected")
    m_ops, a_ops = stencil(mg_op), stencil(outer_op)
    rhs_m = re.search(r"\bField\s+RHS\s*@\s*finest\s+with\s+Complex\s*<\s*Real\s*>\s+on\s+Node\s+of\s+\w+\s*=\s*(.+)$", src, flags=re.M)
    rhs_expr = Expression(rhs_m.group(1)) if rhs_m else Expression("0.0")
    restr = re.search(r"\bOperator\s+(\w+)\s+from\s+default\s+restriction\b", src)
    prol = re.search(r"\bOperator\s+(\w+)\s+from\s+default\s+prolongation\b", src)
    constants = {g: v for g, v in values.items() if g != "k"}

    def env(level, params):
        h = 1.0 / (1 << level)
        return dict(constants, k=params["k"], PI=np.pi, vf_gridWidth_x=h, vf_gridWidth_y=h)

    def make_stencil(entries):
        def st(level, params):
            out = [0.0] * 7
            e = env(level, params)
            for off, expr in entries.items():
                out[_star_slot(off)] = expr.evaluate(e)
            return out
        return st

    def u0(level):
        n = (1 << level) + 1
        return np.zeros((n, n), dtype=np.complex128)

    def rhs(level):
        n = (1 << level) + 1
        x = np.arange(n) / (n - 1)
        h = 1.0 / (n - 1)
        e = dict(constants, k=values["k"], PI=np.pi, vf_gridWidth_x=h, vf_gridWidth_y=h,
                 vf_nodePos_x=x[None, :], vf_nodePos_y=x[:, None])
        return (np.zeros((n, n)) + rhs_expr.evaluate(e)).astype(np.complex128)

    return problems.ProblemDescription(name, 2, min_level, max_level,
                                       scalar_kind=abi.COMPLEX_F64, bc_kind=abi.BC_HELMHOLTZ_ROBIN,
                                       outer_kind=abi.OUTER_PBICGSTAB, cgs_kind=abi.CGS_BICGSTAB,
                                       max_iterations=max_its, cgs_max_iterations=cgs_its,
                                       target_res_reduction=tol, cgs_target_res_reduction=cgs_tol,
                                       field_name=field_name, rhs_name=rhs_name, operator_name=mg_op, equation_name=eq_name,
                                       restriction_name=restr.group(1) if restr else f"Restriction_{field_name}",
                                       prolongation_name=prol.group(1) if prol else f"Prolongation_{field_name}",
                                       mg_stencil=make_stencil(m_ops), outer_stencil=make_stencil(a_ops),
                                       initial_solution=u0, right_hand_side=rhs, parameters={"k": float(np.real(values["k"]))})


def load_problem(base_path: str, settings_path: str, knowledge_path: str,
                 min_level: Optional[int] = None, max_level: Optional[int] = None) -> problems.ProblemDescription:
    """The three constructor arguments of the reference's ProgramGenerator (exastencils.py:40-43) -> descriptor."""
    settings = resolve_settings(read_key_values(os.path.join(base_path, settings_path)))
    knowledge = read_key_values(os.path.join(base_path, knowledge_path))
    name = settings.get("configName")
    if name is None:
        raise UnsupportedProblem("settings file has no configName")
    prefix = os.path.join(base_path, settings.get("basePathPrefix", "."))
    l2_path = os.path.join(prefix, settings.get("l2file", name + ".exa2"))
    if not os.path.exists(l2_path):
        l3_path = os.path.join(prefix, settings.get("l3file", name + ".exa3"))
        l4_path = os.path.join(prefix, settings.get("l4file", name + ".exa4"))
        if os.path.exists(l3_path) and os.path.exists(l4_path):
            with open(l3_path, "r") as f3, open(l4_path, "r") as f4:
                return describe_layer3_helmholtz(f3.read(), f4.read(), name, int(knowledge["dimensionality"]),
                                                 int(knowledge["minLevel"]) if min_level is None else min_level,
                                                 int(knowledge["maxLevel"]) if max_level is None else max_level)
        raise UnsupportedProblem(f"no layer-2 file {l2_path} and no layer-3 / layer-4 pair: nothing to read the problem from "
                                 f"(built-in descriptors: {sorted(problems.BY_NAME)})")
    with open(l2_path, "r") as fh:
        l2 = parse_layer2(fh.read())
    hints = {}
    l3_path = os.path.join(prefix, settings.get("l3file", name + ".exa3"))
    if os.path.exists(l3_path):
        with open(l3_path, "r") as fh:
            hints = parse_solver_hints(fh.read())
    return describe(l2, name, int(knowledge["dimensionality"]),
                    int(knowledge["minLevel"]) if min_level is None else min_level,
                    int(knowledge["maxLevel"]) if max_level is None else max_level, hints)
