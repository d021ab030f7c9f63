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
