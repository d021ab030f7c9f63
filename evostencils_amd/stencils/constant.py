"""Constant-coefficient stencils: an ordered collection of (offset, value) pairs.

Counterpart of evostencils/stencils/constant.py:1-149, reduced to what the backend's lowering reads
(`entries`, `dimension`) plus the few algebra helpers the smoother constructors need.
"""
from typing import Callable, Iterable, Optional, Sequence, Tuple

Offset = Tuple[int, ...]


class Stencil:
    def __init__(self, entries: Iterable[Tuple[Offset, object]], dimension: Optional[int] = None):
        self._entries = tuple((tuple(int(c) for c in off), val) for off, val in entries)
        if dimension is None:
            if not self._entries:
                raise ValueError("dimension required for an empty stencil")
            dimension = len(self._entries[0][0])
        self._dimension = int(dimension)

    @property
    def entries(self):
        return self._entries

    @property
    def dimension(self) -> int:
        return self._dimension

    @property
    def number_of_entries(self) -> int:
        return len(self._entries)

    def value_at(self, offset: Sequence[int], default=0.0):
        key = tuple(offset)
        for off, val in self._entries:
            if off == key:
                return val
        return default

    def __repr__(self):
        return f"Stencil({self._entries!r})"


def keep(stencil: Optional[Stencil], predicate: Callable[[Offset, object], bool]) -> Optional[Stencil]:
    """Sub-stencil of the entries satisfying `predicate` (reference: constant.py:47-56 filter_stencil)."""
    if stencil is None:
        return None
    return Stencil([(o, v) for o, v in stencil.entries if predicate(o, v)], stencil.dimension)


filter_stencil = keep


def diagonal(stencil: Optional[Stencil]) -> Optional[Stencil]:
    return keep(stencil, lambda o, _: all(c == 0 for c in o))


def scale(factor, stencil: Optional[Stencil]) -> Optional[Stencil]:
    if stencil is None:
        return None
    return Stencil([(o, factor * v) for o, v in stencil.entries], stencil.dimension)


def combine(a: Optional[Stencil], b: Optional[Stencil], f) -> Optional[Stencil]:
    """Entry-wise f(a, b) on the union of the offsets (reference: constant.py add / sub); offsets keep the order of
    first appearance."""
    if a is None or b is None:
        return None
    values, order = {}, []
    for off, val in a.entries:
        values[off] = val
        order.append(off)
    for off, val in b.entries:
        if off in values:
            values[off] = f(values[off], val)
        else:
            values[off] = f(0.0, val)
            order.append(off)
    return Stencil([(o, values[o]) for o in order], a.dimension)


def add(a, b):
    return combine(a, b, lambda x, y: x + y)


def sub(a, b):
    return combine(a, b, lambda x, y: x - y)


def mul(a: Optional[Stencil], b: Optional[Stencil]) -> Optional[Stencil]:
    """Stencil of the operator product a * b (offsets add, values multiply)."""
    if a is None or b is None:
        return None
    values, order = {}, []
    for oa, va in a.entries:
        for ob, vb in b.entries:
            off = tuple(x + y for x, y in zip(oa, ob))
            if off not in values:
                values[off] = 0.0
                order.append(off)
            values[off] = values[off] + va * vb
    return Stencil([(o, values[o]) for o in order], a.dimension)


def get_unit_stencil(grid) -> Stencil:
    return Stencil([((0,) * grid.dimension, 1.0)], grid.dimension)


def get_null_stencil(grid) -> Stencil:
    return Stencil((), grid.dimension)


def is_star(stencil: Stencil) -> bool:
    """True if every offset has at most one non-zero component of magnitude 1 (5-/7-point shapes)."""
    for off, _ in stencil.entries:
        nz = [c for c in off if c != 0]
        if len(nz) > 1 or any(abs(c) != 1 for c in nz):
            return False
    return True
