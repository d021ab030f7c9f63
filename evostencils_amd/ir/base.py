"""Scalar-equation IR nodes (reference: evostencils/ir/base.py).

The node *vocabulary* (class names, constructor argument order, attribute names) follows the
reference because it is the input format of the path this package replaces; the implementation is
a fresh, much smaller one: plain attribute holders, no LFA hooks, no stencil-generator plumbing
beyond `generate_stencil()`.
"""
from functools import reduce
import operator as _op

from evostencils_amd.ir import partitioning as part
from evostencils_amd.stencils import constant, multiple


def _npoints(grid):
    return reduce(_op.mul, grid.size)


class Expression:
    """Base of all nodes. `valid` marks shared sub-expressions already emitted by a lowering pass
    (reference: base.py:9-13, used at exastencils.py:699-702,746-749,828-830)."""

    def __init__(self):
        self.valid = False
        self.lfa_symbol = None
        self.runtime = None

    def mutate(self, f, *args):  # visit children that a lowering pass may have flagged
        pass


class Grid:
    """size = cells per dimension (2^level), spacing = mesh width, level (reference: base.py:168-196)."""

    def __init__(self, size, spacing, level):
        if len(size) != len(spacing):
            raise ValueError("Dimensions of the size and step size must match")
        self._size, self._spacing, self._level = tuple(size), tuple(spacing), level

    size = property(lambda self: self._size)
    spacing = property(lambda self: self._spacing)
    level = property(lambda self: self._level)
    dimension = property(lambda self: len(self._size))

    def __eq__(self, other):
        return isinstance(other, Grid) and self.size == other.size and self.spacing == other.spacing

    def __hash__(self):
        return hash((self._size, self._spacing))

    def __repr__(self):
        return f"Grid({self._size!r}, {self._spacing!r}, {self._level!r})"


class ConstantStencilGenerator:
    def __init__(self, stencil):
        self._stencil = stencil

    def generate_stencil(self, _grid=None):
        return self._stencil


class Entity(Expression):
    def __init__(self, name, grid, shape):
        super().__init__()
        self._name, self._grid, self._shape = name, grid, shape

    name = property(lambda self: self._name)
    grid = property(lambda self: self._grid)
    shape = property(lambda self: self._shape)

    def __str__(self):
        return str(self._name)


class Operator(Entity):
    def __init__(self, name, grid, stencil_generator=None):
        n = _npoints(grid)
        super().__init__(name, grid, (n, n))
        self._stencil_generator = stencil_generator

    stencil_generator = property(lambda self: self._stencil_generator)

    def generate_stencil(self):
        if self._stencil_generator is None:
            return None
        return self._stencil_generator.generate_stencil(self._grid)

    def __repr__(self):
        return f"{type(self).__name__}({self._name!r}, {self._grid!r})"


class Identity(Operator):
    def __init__(self, grid, name="I"):
        super().__init__(name, grid, ConstantStencilGenerator(constant.get_unit_stencil(grid)))


class ZeroOperator(Operator):
    def __init__(self, grid, shape=None, name="0"):
        super().__init__(name, grid, ConstantStencilGenerator(constant.get_null_stencil(grid)))
        if shape is not None:
            self._shape = shape


class Approximation(Entity):
    def __init__(self, name, grid):
        super().__init__(name, grid, (_npoints(grid), 1))

    predecessor = property(lambda self: None)

    def __repr__(self):
        return f"{type(self).__name__}({self._name!r}, {self._grid!r})"


class RightHandSide(Approximation):
    pass


class ZeroApproximation(Approximation):
    def __init__(self, grid, name="0"):
        super().__init__(name, grid)


class UnaryExpression(Expression):
    def __init__(self, operand):
        super().__init__()
        self._operand = operand

    operand = property(lambda self: self._operand)
    shape = property(lambda self: self._operand.shape)
    grid = property(lambda self: self._operand.grid)

    def mutate(self, f, *args):
        f(self._operand, *args)

    def __repr__(self):
        return f"{type(self).__name__}({self._operand!r})"


class Inverse(UnaryExpression):
    def __str__(self):
        return f"{self._operand}.I"


class Diagonal(UnaryExpression):
    def generate_stencil(self):
        return multiple.diagonal(self._operand.generate_stencil())


class BinaryExpression(Expression):
    def __init__(self, operand1, operand2):
        super().__init__()
        self._operand1, self._operand2 = operand1, operand2

    operand1 = property(lambda self: self._operand1)
    operand2 = property(lambda self: self._operand2)
    grid = property(lambda self: self._operand1.grid)

    def mutate(self, f, *args):
        f(self._operand1, *args)
        f(self._operand2, *args)

    def __repr__(self):
        return f"{type(self).__name__}({self._operand1!r}, {self._operand2!r})"


class Addition(BinaryExpression):
    shape = property(lambda self: self._operand1.shape)

    def generate_stencil(self):
        return constant.add(self._operand1.generate_stencil(), self._operand2.generate_stencil())


class Subtraction(BinaryExpression):
    shape = property(lambda self: self._operand1.shape)

    def generate_stencil(self):
        return constant.sub(self._operand1.generate_stencil(), self._operand2.generate_stencil())


class Multiplication(BinaryExpression):
    def __init__(self, operand1, operand2):
        if operand1.shape[1] != operand2.shape[0]:
            raise AssertionError("Operand shapes are not aligned")
        super().__init__(operand1, operand2)

    shape = property(lambda self: (self._operand1.shape[0], self._operand2.shape[1]))

    def generate_stencil(self):
        """Operator product (only meaningful when both operands are operators: the row entries of a system of
        PDEs, grammar/multigrid.py _term_to_ir)."""
        return constant.mul(self._operand1.generate_stencil(), self._operand2.generate_stencil())


class Scaling(Expression):
    def __init__(self, factor, operand):
        super().__init__()
        self._factor, self._operand = factor, operand

    factor = property(lambda self: self._factor)
    operand = property(lambda self: self._operand)
    grid = property(lambda self: self._operand.grid)
    shape = property(lambda self: self._operand.shape)

    def mutate(self, f, *args):
        f(self._operand, *args)

    def generate_stencil(self):
        return constant.scale(self._factor, self._operand.generate_stencil())


class InterGridOperator(Operator):
    def __init__(self, name, grid, fine_grid, coarse_grid, stencil_generator):
        super().__init__(name, grid, stencil_generator)
        self._fine_grid, self._coarse_grid = fine_grid, coarse_grid

    fine_grid = property(lambda self: self._fine_grid)
    coarse_grid = property(lambda self: self._coarse_grid)


class Restriction(InterGridOperator):
    """Maps fine -> coarse; lives on the coarse grid (reference: base.py:502-529)."""

    def __init__(self, name, fine_grid, coarse_grid, stencil_generator=None):
        super().__init__(name, coarse_grid, fine_grid, coarse_grid, stencil_generator)
        self._shape = (_npoints(coarse_grid), _npoints(fine_grid))


class ZeroRestriction(Restriction):
    def __init__(self, fine_grid, coarse_grid, name="0"):
        super().__init__(name, fine_grid, coarse_grid, None)


class Prolongation(InterGridOperator):
    """Maps coarse -> fine; lives on the fine grid (reference: base.py:537-564)."""

    def __init__(self, name, fine_grid, coarse_grid, stencil_generator=None):
        super().__init__(name, fine_grid, fine_grid, coarse_grid, stencil_generator)
        self._shape = (_npoints(fine_grid), _npoints(coarse_grid))


class ZeroProlongation(Prolongation):
    def __init__(self, fine_grid, coarse_grid, name="0"):
        super().__init__(name, fine_grid, coarse_grid, None)


class CoarseGridSolver(Entity):
    def __init__(self, name, operator, expression=None):
        super().__init__(name, operator.grid, operator.shape)
        self._operator, self._expression = operator, expression

    operator = property(lambda self: self._operator)
    expression = property(lambda self: self._expression)

    def mutate(self, f, *args):
        f(self._operator, *args)


class Residual(Expression):
    """rhs - operator * approximation (reference: base.py:598-648)."""

    def __init__(self, operator, approximation, rhs):
        super().__init__()
        self._operator, self._approximation, self._rhs = operator, approximation, rhs

    operator = property(lambda self: self._operator)
    approximation = property(lambda self: self._approximation)
    rhs = property(lambda self: self._rhs)
    shape = property(lambda self: self._rhs.shape)
    grid = property(lambda self: self._rhs.grid)

    def mutate(self, f, *args):
        f(self._rhs, *args)
        f(self._approximation, *args)

    def __repr__(self):
        return f"Residual({self._operator!r}, {self._approximation!r}, {self._rhs!r})"


class Cycle(Expression):
    """approximation + relaxation_factor * correction; mutable on purpose: the grammar's productions
    rewrite `correction`, `relaxation_factor`, `partitioning` in place (reference: base.py:651-697,
    grammar/multigrid.py:257-267,283)."""

    def __init__(self, approximation, rhs, correction=None, partitioning=part.Single,
                 relaxation_factor=1.0, predecessor=None):
        super().__init__()
        self.approximation = approximation
        self.rhs = rhs
        self.correction = correction
        self.partitioning = partitioning
        self.relaxation_factor = relaxation_factor
        self.predecessor = predecessor
        self.global_id = None

    shape = property(lambda self: self.approximation.shape)
    grid = property(lambda self: self.approximation.grid)

    def mutate(self, f, *args):
        f(self.correction, *args)

    def __repr__(self):
        return (f"Cycle({self.approximation!r}, {self.rhs!r}, {self.correction!r}, "
                f"{self.partitioning!r}, {self.relaxation_factor!r})")


def get_coarse_grid(grid, coarsening_factor):
    return Grid(tuple(s // c for s, c in zip(grid.size, coarsening_factor)),
                tuple(h * c for h, c in zip(grid.spacing, coarsening_factor)),
                grid.level - 1)


def get_coarse_operator(operator, coarse_grid):
    return Operator(operator.name, coarse_grid, operator.stencil_generator)


def add(a, b):
    return Addition(a, b)


def sub(a, b):
    return Subtraction(a, b)


def mul(a, b):
    return Multiplication(a, b)


def inv(a):
    return Inverse(a)
