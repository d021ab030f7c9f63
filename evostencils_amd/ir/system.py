"""Block (multi-field) wrappers around the scalar IR nodes (reference: evostencils/ir/system.py).
For the scalar PDEs in scope every wrapper holds exactly one entry, but the lowering walks
`entries` so the structure is kept."""
from evostencils_amd.ir import base


class System(base.Expression):
    def __init__(self, name, entries, shape):
        super().__init__()
        self._name, self._entries, self._shape = name, entries, shape

    name = property(lambda self: self._name)
    entries = property(lambda self: self._entries)
    shape = property(lambda self: self._shape)


class Operator(System):
    def __init__(self, name, entries):
        rows = sum(row[0].shape[0] for row in entries)
        cols = sum(e.shape[1] for e in entries[0])
        super().__init__(name, entries, (rows, cols))

    @property
    def grid(self):
        return [e.grid for e in self._entries[0]]


class Approximation(System):
    def __init__(self, name, entries):
        shape = (sum(e.shape[0] for e in entries), entries[0].shape[1])
        super().__init__(name, entries, shape)

    @property
    def grid(self):
        return [e.grid for e in self._entries]

    predecessor = property(lambda self: None)


class RightHandSide(Approximation):
    pass


class ZeroApproximation(Approximation):
    def __init__(self, grid, name="0"):
        super().__init__(name, [base.ZeroApproximation(g) for g in grid])


class InterGridOperator(Operator):
    def __init__(self, name, list_of_intergrid_operators, zero_type):
        ops = list_of_intergrid_operators
        entries = [[op if i == j else zero_type(op.fine_grid, op.coarse_grid) for j in range(len(ops))]
                   for i, op in enumerate(ops)]
        super().__init__(name, entries)


class Restriction(InterGridOperator):
    def __init__(self, name, list_of_intergrid_operators):
        super().__init__(name, list_of_intergrid_operators, base.ZeroRestriction)


class Prolongation(InterGridOperator):
    def __init__(self, name, list_of_intergrid_operators):
        super().__init__(name, list_of_intergrid_operators, base.ZeroProlongation)


class Diagonal(base.UnaryExpression):
    """Decoupled point smoother: per-equation diagonal (reference: system.py:123-124)."""


class ElementwiseDiagonal(base.UnaryExpression):
    """Collective point smoother (reference: system.py:127-129)."""

    def __str__(self):
        return "D"


class Jacobian(base.UnaryExpression):
    """Marker for the derivative of the non-linear term in a Newton-type point smoother; carries the number of local
    Newton steps (reference: system.py:132-138)."""

    def __init__(self, dummy_operator, n_newton_steps):
        self.n_newton_steps = n_newton_steps
        super().__init__(dummy_operator)

    def __str__(self):
        return f"J[{self.n_newton_steps}]"


def get_coarse_grid(grid, coarsening_factors):
    return [base.get_coarse_grid(g, cf) for g, cf in zip(grid, coarsening_factors)]


def get_coarse_approximation(approximation, coarsening_factors):
    return Approximation(approximation.name,
                         [base.Approximation(f"{e.name}_c", base.get_coarse_grid(e.grid, cf))
                          for e, cf in zip(approximation.entries, coarsening_factors)])


def get_coarse_rhs(rhs, coarsening_factors):
    return RightHandSide(rhs.name,
                         [base.RightHandSide(f"{e.name}_c", base.get_coarse_grid(e.grid, cf))
                          for e, cf in zip(rhs.entries, coarsening_factors)])
