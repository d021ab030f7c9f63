"""Periodic ("multiple") stencils: a d-dimensional nested tuple of constant stencils, one per
position inside the period. Counterpart of evostencils/stencils/multiple.py:1-240, reduced to the
block-diagonal construction that block smoothers are made of.

Nesting order = coordinate order: the outermost tuple runs over i0 (x), as in the reference
(multiple.py:155-169 indexed_map_stencil).
"""
from itertools import product
from typing import Sequence, Tuple

from evostencils_amd.stencils import constant


class Stencil:
    def __init__(self, constant_stencils, dimension: int):
        if dimension < 1:
            raise ValueError("dimension must be >= 1")
        self._constant_stencils = constant_stencils
        self._dimension = dimension

    @property
    def constant_stencils(self):
        return self._constant_stencils

    @property
    def dimension(self) -> int:
        return self._dimension


def period_shape(stencil) -> Tuple[int, ...]:
    """Extent of the period per dimension (reference: multiple.py:103-117 determine_maximal_shape).
    A constant stencil has period (1,)*d."""
    if isinstance(stencil, constant.Stencil) or not hasattr(stencil, "constant_stencils"):
        return (1,) * stencil.dimension
    shape = []
    level = stencil.constant_stencils
    for _ in range(stencil.dimension):
        shape.append(len(level))
        level = level[0]
    return tuple(shape)


determine_maximal_shape = period_shape


def at(stencil, index: Sequence[int]):
    """Constant stencil at period position `index` (wraps around)."""
    if not hasattr(stencil, "constant_stencils"):
        return stencil
    node = stencil.constant_stencils
    for i in index:
        node = node[i % len(node)]
    return node


def _build(shape: Sequence[int], f):
    def rec(prefix, rest):
        if not rest:
            return f(tuple(prefix))
        return tuple(rec(prefix + [i], rest[1:]) for i in range(rest[0]))
    return rec([], list(shape))


def block_diagonal(stencil, block_size: Sequence[int]) -> Stencil:
    """Restriction of a constant stencil to non-overlapping blocks of shape `block_size`: at block
    cell `index` only the entries whose target `index + offset` stays inside the block survive
    (reference: multiple.py:204-217)."""
    block_size = tuple(int(b) for b in block_size)
    if len(block_size) != stencil.dimension:
        raise ValueError("Block size does not match dimension of the problem")

    def cell(index):
        base = at(stencil, index)
        return constant.keep(base, lambda o, _: all(0 <= i + c < b for i, c, b in zip(index, o, block_size)))

    return Stencil(_build(block_size, cell), stencil.dimension)


def diagonal(stencil):
    if not hasattr(stencil, "constant_stencils"):
        return constant.diagonal(stencil)
    shape = period_shape(stencil)
    return Stencil(_build(shape, lambda idx: constant.diagonal(at(stencil, idx))), stencil.dimension)


def is_block_diagonal_of(candidate, operator_stencil, block_size) -> bool:
    """True if `candidate` equals block_diagonal(operator_stencil, block_size) entry for entry."""
    expected = block_diagonal(operator_stencil, block_size)
    for idx in product(*[range(b) for b in block_size]):
        a = sorted(((o, complex(v)) for o, v in at(candidate, idx).entries), key=lambda t: t[0])
        b = sorted(((o, complex(v)) for o, v in at(expected, idx).entries), key=lambda t: t[0])
        if len(a) != len(b):
            return False
        for (oa, va), (ob, vb) in zip(a, b):
            if oa != ob or va != vb:
                return False
    return True
