"""Point partitionings of a smoothing sweep (reference: evostencils/ir/partitioning.py:18-47).

Only the two partitionings the reference defines exist: ``Single`` (one Jacobi-type sweep) and
``RedBlack`` (two coloured half sweeps, colour = (i0+i1[+i2]) % 2).
The classes themselves (not instances) are used as values, as in the reference.
"""


class Single:
    @staticmethod
    def get_name():
        return "single"

    def __repr__(self):
        return "Single()"


class RedBlack:
    @staticmethod
    def get_name():
        return "red_black"

    def __repr__(self):
        return "RedBlack()"


def kind_of(partitioning) -> str:
    """'single' | 'red_black' for a partitioning class or instance (ours or the reference's)."""
    name = getattr(partitioning, "__name__", None) or type(partitioning).__name__
    if name == "Single":
        return "single"
    if name == "RedBlack":
        return "red_black"
    raise ValueError(f"unsupported partitioning {partitioning!r}")
