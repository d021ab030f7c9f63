"""Seeded generator of candidate cycles for population benchmarks and tests ("256 candidate op
programs from a seeded generator drawing per level: smoother in {Jacobi, RB-GS, block-Jacobi(2,1),(1,2),(2,2)},
nu_pre, nu_post in {0,1,2,3}, omega in linspace(0.1,1.9,37), coarse-grid-correction omega likewise").

Candidates are built with the grammar's own productions (evostencils_amd.grammar.productions), so each one is a DAG
the reference's G3P search could have produced and goes through the same lowering as an evolved individual."""
from typing import List

import numpy as np

SMOOTHERS_2D = ["jacobi", "rbgs", ("block", (2, 1)), ("block", (1, 2)), ("block", (2, 2))]
SMOOTHERS_3D = ["jacobi", "rbgs", ("block", (2, 1, 1)), ("block", (1, 2, 1)), ("block", (2, 2, 1))]


def _smooth(ctx, depth, cycle, kind, omega_index):
    if kind == "jacobi":
        return ctx[f"collective_jacobi_{depth}"](omega_index, ctx["single"], cycle)
    if kind == "rbgs":
        return ctx[f"collective_jacobi_{depth}"](omega_index, ctx["red_black"], cycle)
    return ctx[f"collective_block_jacobi_{depth}"](omega_index, (tuple(kind[1]),), cycle)


def random_v_cycle(ctx, depth: int, dimension: int, rng: np.random.Generator, stable_only: bool = True):
    """One random V-type candidate over `depth`+1 levels. With `stable_only` the relaxation factors are drawn from the
    range in which the smoother converges (Jacobi/block: <= 1.0, RB-GS: <= 1.5), so most candidates are usable solvers;
    otherwise from the grammar's full set np.linspace(0.1, 1.9, 37)."""
    smoothers = SMOOTHERS_2D if dimension == 2 else SMOOTHERS_3D

    def draw():
        kind = smoothers[int(rng.integers(len(smoothers)))]
        if stable_only:
            hi = 28 if kind == "rbgs" else 18          # indices of 1.5 and 1.0
            idx = int(rng.integers(6, hi + 1))          # >= 0.4
        else:
            idx = int(rng.integers(0, 37))
        return kind, idx

    pre = [[draw() for _ in range(int(rng.integers(0, 4)))] for _ in range(depth)]
    post = [[draw() for _ in range(int(rng.integers(0, 4)))] for _ in range(depth)]
    cgc = [int(rng.integers(12, 23)) if stable_only else int(rng.integers(0, 37)) for _ in range(depth)]
    state = ctx["u_and_f"]
    cycle = ctx["residual_0"](state)
    for d in range(depth):
        for kind, idx in pre[d]:
            state = _smooth(ctx, d, cycle, kind, idx)
            cycle = ctx[f"residual_{d}"](state)
        if d < depth - 1:
            cycle = ctx[f"coarsening_{d}"](ctx[f"A_{d + 1}"], ctx[f"zero_{d + 1}"], ctx[f"R_{d}"], cycle)
        else:
            state = ctx[f"correct_with_coarse_grid_solver_{d}"](cgc[d], ctx[f"P_{d + 1}"], ctx[f"CGS_{d + 1}"],
                                                                ctx[f"R_{d}"], cycle)
    for d in range(depth - 1, -1, -1):
        if d < depth - 1:
            state = ctx[f"update_with_coarse_grid_correction_{d}"](cgc[d], ctx[f"P_{d + 1}"], state)
        for kind, idx in post[d]:
            cycle = ctx[f"residual_{d}"](state)
            state = _smooth(ctx, d, cycle, kind, idx)
    return state[0]


def population(make_context, depth: int, dimension: int, size: int, seed: int = 2024, stable_only: bool = True) -> List:
    """`size` candidates; `make_context()` must return a FRESH production context per candidate (the productions mutate
    and share DAG nodes, one context = one individual)."""
    rng = np.random.default_rng(seed)
    return [random_v_cycle(make_context(), depth, dimension, rng, stable_only) for _ in range(size)]
