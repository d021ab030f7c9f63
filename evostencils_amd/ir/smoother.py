"""Smoother constructors: map a system operator to the operator whose inverse is applied to the
residual (reference: evostencils/ir/smoother.py:5-24)."""
from evostencils_amd.ir import base, system
from evostencils_amd.stencils import multiple


def generate_decoupled_jacobi(operator):
    return system.Diagonal(operator)


def generate_collective_jacobi(operator):
    return system.ElementwiseDiagonal(operator)


def generate_collective_block_jacobi(operator, block_sizes):
    """`block_sizes`: one block shape per field, e.g. ((2, 1),)."""
    rows = []
    for i, row in enumerate(operator.entries):
        new_row = []
        for j, entry in enumerate(row):
            blocked = multiple.block_diagonal(entry.generate_stencil(), block_sizes[i])
            new_row.append(base.Operator(f"{operator.name}_{i}{j}_block_diag", entry.grid,
                                         base.ConstantStencilGenerator(blocked)))
        rows.append(new_row)
    return system.Operator(f"{operator.name}_block_diag", rows)


def generate_jacobi_picard(operator):
    """Non-linear FAS: point smoother that divides by the linear operator's diagonal only (smoother.py:41-42)."""
    return system.ElementwiseDiagonal(operator)


def generate_jacobi_newton(operator, n_newton_steps):
    """Non-linear FAS: diagonal plus the derivative of the non-linear term, `n_newton_steps` local steps
    (smoother.py:45-46)."""
    return base.Addition(system.ElementwiseDiagonal(operator), system.Jacobian(operator, n_newton_steps))
