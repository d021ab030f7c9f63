"""Statement list -> ExaSlang layer-3 text, character for character what the reference's
``generate_multigrid`` / ``generate_cycle_function`` print (exastencils.py:318-336,608-633,659-667,
684-925). Pinned by golden strings captured from the reference (tests/golden/).

sympy is used exactly where the reference uses it (local equations, operator products), so term
ordering and float formatting are sympy's in both.
"""
from itertools import product

from evostencils_amd.code_generation import lowering as low


class FieldNames:
    """Field naming of CycleStorage (exastencils.py:14-26): solution `u`, rhs from the equation,
    residual `gen_residual_u`, correction `gen_error_u`; coarse-solver rhs `gen_rhs_u`."""

    def __init__(self, field_names, rhs_names):
        self.field_names = list(field_names)
        self.rhs_names = list(rhs_names)

    def exa(self, ref: low.FieldRef) -> str:
        f = self.field_names[ref.index]
        if ref.role == low.SOLUTION:
            name = f
        elif ref.role == low.RHS:
            name = self.rhs_names[ref.index]
        elif ref.role == low.RESIDUAL:
            name = f"gen_residual_{f}"
        elif ref.role == low.CORRECTION:
            name = f"gen_error_{f}"
        elif ref.role == low.CGS_RHS:
            name = f"gen_rhs_{f}"
        else:
            raise ValueError(ref.role)
        return f"{name}@{ref.level}"


def _operator_product(entry, level, field_exa):
    """` - (Laplace@L*u@L)` term via sympy MatrixSymbols (exastencils.py:635-657,723-725,850-852)."""
    import sympy

    def rec(e):
        if low.isa(e, "base", "Multiplication"):
            return rec(e.operand1) * rec(e.operand2)
        if low.isa(e, "base", "Addition"):
            return rec(e.operand1) + rec(e.operand2)
        if low.isa(e, "base", "Subtraction"):
            return rec(e.operand1) - rec(e.operand2)
        if low.isa(e, "base", "Scaling"):
            return e.factor * rec(e.operand)
        if low.isa(e, "base", "Identity"):
            return sympy.sympify(1)
        if low.isa(e, "base", "Operator"):
            return sympy.MatrixSymbol(f"{e.name}@{level}", e.shape[0], e.shape[1])
        raise RuntimeError("Invalid expression")

    expr = rec(entry) * sympy.MatrixSymbol(field_exa, entry.shape[1], entry.shape[1])
    return f" - ({expr.expand()})"


def _operator_terms(entry, level, field_exa):
    if low.isa(entry, "base", "Identity"):
        return field_exa
    if low.isa(entry, "base", "ZeroOperator"):
        return ""
    return _operator_product(entry, level, field_exa)


def _system_local_equations(st: low.Statement, names: FieldNames, max_level: int):
    """Systems of PDEs: for every unknown of the block (field order) and every block cell (i0 outermost) the line
    `x_i@[c] => (sum_j A_ij x_j) == f_i@[c]` (ir/transformations.py:51-121: dictionary order = fields, then cells)."""
    import sympy
    level = st.level
    prefix = "gen_error_" if level < max_level else ""
    lines = []
    for k, i in enumerate(st.unknowns):
        for idx in product(*[range(b) for b in (st.blocks[k] if st.blocks else st.block)]):
            expr = sympy.sympify(0)
            for j, stencil in enumerate(st.system_stencils[i]):
                for off, val in stencil.entries:
                    pos = ", ".join(str(int(c) + int(o)) for c, o in zip(idx, off))
                    expr = expr + val * sympy.Symbol(f"{prefix}{names.field_names[j]}@{level}@[{pos}]")
            here = ", ".join(str(c) for c in idx)
            lines.append((f"{prefix}{names.field_names[i]}@{level}@[{here}]", str(expr),
                          f"{names.rhs_names[i]}@{level}@[{here}]"))
    return lines


def _local_equations(st: low.Statement, names: FieldNames, max_level: int):
    """One `unknown => (row of A) == rhs` line per block cell, i0 outermost
    (ir/transformations.py:51-121 + exastencils.py:608-633)."""
    import sympy
    if st.unknowns:
        return _system_local_equations(st, names, max_level)
    fname = names.field_names[st.dst.index]
    rhs_name = names.rhs_names[st.dst.index]
    level = st.level
    prefix = "gen_error_" if level < max_level else ""
    lines = []
    for idx in product(*[range(b) for b in st.block]):
        expr = sympy.sympify(0)
        for off, val in st.operator_stencil.entries:
            pos = ", ".join(str(int(i) + int(o)) for i, o in zip(idx, off))
            expr = expr + val * sympy.Symbol(f"{prefix}{fname}@{level}@[{pos}]")
        here = ", ".join(str(i) for i in idx)
        lines.append((f"{prefix}{fname}@{level}@[{here}]", str(expr), f"{rhs_name}@{level}@[{here}]"))
    return lines


def _coloring(dimension, indentation):
    return f"{indentation}\t((" + " + ".join(f"i{i}" for i in range(dimension)) + ") % 2),\n"


def render_statement(st: low.Statement, names: FieldNames, max_level: int, dimension: int,
                     use_jacobi_prefix: bool = True) -> str:
    k = st.kind
    if k == "zero":
        return f"\t{names.exa(st.dst)} = 0\n"
    if k == "copy":
        return f"\t{names.exa(st.dst)} = {names.exa(st.src)}\n"
    if k == "coarse_solve":
        return f"\t{st.cycle_name}@{st.level}()\n"
    if k == "residual":
        if st.operator_row is not None:
            terms = "".join(_operator_terms(entry, st.level, names.exa(fld))
                            for entry, fld in zip(st.operator_row, st.row_fields))
            return f"\t{names.exa(st.dst)} = {names.exa(st.rhs)}{terms}\n"
        return (f"\t{names.exa(st.dst)} = {names.exa(st.rhs)}"
                f"{_operator_terms(st.operator_entry, st.level, names.exa(st.src))}\n")
    if k == "richardson":
        return (f"\t{names.exa(st.dst)} += {st.weight} * ({names.exa(st.rhs)}"
                f"{_operator_terms(st.operator_entry, st.level, names.exa(st.dst))})\n")
    if k in ("prolong_add", "restrict_add"):
        return (f"\t{names.exa(st.dst)} += {st.weight} * ({st.operator_name}@{st.operator_level} * "
                f"{names.exa(st.src)})\n")
    if k in ("restrict", "prolong_set"):
        return f"\t{names.exa(st.dst)} = {st.operator_name}@{st.operator_level} * {names.exa(st.src)}\n"
    if k == "smooth":
        coloring = st.partitioning != "single"
        jacobi_prefix = "with jacobi " if (use_jacobi_prefix and not coloring) else ""
        ind = "\t" if coloring else ""
        text = ""
        if coloring:
            text += f"{ind}color with {{\n" + _coloring(dimension, ind)
        # `solve locally at <field>@L`: the reference names the field without the @[..] access
        head = names.exa(st.dst)
        text += f"\t{ind}solve locally at {head} {jacobi_prefix}relax {st.weight} {{\n"
        for unknown, row, rhs in _local_equations(st, names, max_level):
            text += f"\t\t{ind}{unknown} => ({row}) == {rhs}\n"
        text += f"\t{ind}}}\n"
        if coloring:
            text += "\t}\n"
        return text
    raise ValueError(k)


def render(statements, names: FieldNames, max_level: int, dimension: int, use_jacobi_prefix=True) -> str:
    return "".join(render_statement(s, names, max_level, dimension, use_jacobi_prefix) for s in statements)
