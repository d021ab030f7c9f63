"""Candidate cycles used by the tests, written against a *context* (name -> production/terminal
dictionary). The same functions run against evostencils_amd.grammar.productions.generate_context
(our IR) and, in tests/golden/generate_golden.py, against the reference's own productions captured
from evostencils.grammar.multigrid.add_level — so both sides build the identical DAG shape.

Relaxation factors are indices into np.linspace(0.1, 1.9, 37) (grammar/multigrid.py:428):
index i -> 0.1 + 0.05 i, e.g. 14 -> 0.8, 18 -> 0.9999999999999999, 21 -> 1.15.
"""


from evostencils_amd.candidates import smooth, v_cycle      # noqa: F401 — the cases below and the tests use them


def tutorial_cycle(P):
    """The evolved 5-level cycle recorded in the reference's tutorial notebook
    (notebooks/tutorial.ipynb:3422-3470): levels 9..5, depth 4."""
    c = P["residual_0"](P["u_and_f"])
    c = P["coarsening_0"](P["A_1"], P["zero_1"], P["R_0"], c)
    c = P["coarsening_1"](P["A_2"], P["zero_2"], P["R_1"], c)
    c = P["coarsening_2"](P["A_3"], P["zero_3"], P["R_2"], c)
    s = P["collective_block_jacobi_3"](5, ((3, 2),), c)            # 0.35
    s = P["collective_block_jacobi_3"](0, ((1, 6),), P["residual_3"](s))   # 0.1
    s = P["correct_with_coarse_grid_solver_3"](0, P["P_4"], P["CGS_4"], P["R_3"], P["residual_3"](s))  # 0.1
    s = P["update_with_coarse_grid_correction_2"](31, P["P_3"], s)  # 1.65
    s = P["update_with_coarse_grid_correction_1"](33, P["P_2"], s)  # 1.75
    s = P["collective_jacobi_1"](13, P["single"], P["residual_1"](s))   # 0.7499999999999999
    s = P["update_with_coarse_grid_correction_0"](18, P["P_1"], s)  # 0.9999999999999999
    s = P["collective_jacobi_0"](31, P["single"], P["residual_0"](s))   # 1.65
    s = P["collective_jacobi_0"](28, P["red_black"], P["residual_0"](s))  # 1.5
    return s


def elasticity_mixed_cycle(P):
    """Every smoother kind the grammar has for systems of PDEs, three levels: collective red-black, decoupled Jacobi,
    collective block-Jacobi (2,1) on the coarser level, decoupled red-black."""
    c = P["residual_0"](P["u_and_f"])
    s = P["collective_jacobi_0"](14, P["red_black"], c)
    s = P["decoupled_jacobi_0"](10, P["single"], P["residual_0"](s))
    c = P["residual_0"](s)
    c = P["coarsening_0"](P["A_1"], P["zero_1"], P["R_0"], c)
    s = P["collective_block_jacobi_1"](12, ((2, 1), (2, 1)), c)
    s = P["correct_with_coarse_grid_solver_1"](18, P["P_2"], P["CGS_2"], P["R_1"], P["residual_1"](s))
    s = P["update_with_coarse_grid_correction_0"](18, P["P_1"], s)
    s = P["decoupled_jacobi_0"](12, P["red_black"], P["residual_0"](s))
    return s


# name -> (problem, min_level, max_level, builder(P))
def _jac(i): return ("jacobi", i)
def _rb(i): return ("rbgs", i)
def _blk(i, shape): return ("block", i, shape)


CASES = {
    # C1: 3-level V(1,1) Jacobi w=0.8 on 2D Poisson level 8
    "poisson2d_L8_v11_jacobi": ("2D_FD_Poisson_fromL2", 6, 8,
                                lambda P: v_cycle(P, 2, [[_jac(14)], [_jac(14)]], [[_jac(14)], [_jac(14)]])),
    # fine Jacobi, coarse RB-GS (SURVEY App. E probe)
    "poisson2d_L8_v11_mixed": ("2D_FD_Poisson_fromL2", 6, 8,
                               lambda P: v_cycle(P, 2, [[_jac(14)], [_rb(18)]], [[_jac(14)], [_rb(18)]])),
    # 4-level V(2,1) RB-GS w=1.15 (the reference's default smoother setting, Poisson exa3:6-10)
    "poisson2d_L7_v21_rbgs": ("2D_FD_Poisson_fromL2", 4, 7,
                              lambda P: v_cycle(P, 3, [[_rb(21)] * 2] * 3, [[_rb(21)]] * 3)),
    # block smoothers of several shapes, irregular sweep counts
    "poisson2d_L7_blocks": ("2D_FD_Poisson_fromL2", 4, 7,
                            lambda P: v_cycle(P, 3, [[_blk(10, (2, 1))], [_blk(12, (2, 2)), _jac(9)], []],
                                              [[_blk(16, (1, 2)), _rb(20)], [], [_blk(5, (3, 2))]], cgc_index=17)),
    "poisson2d_tutorial": ("2D_FD_Poisson_fromL2", 5, 9, tutorial_cycle),
    # two complete V(1,1) corrections per cycle (irregular shapes of this kind are what the grammar evolves)
    "poisson2d_L7_double_v": ("2D_FD_Poisson_fromL2", 4, 7,
                              lambda P: v_cycle(P, 3, [[_rb(20)]] * 3, [[_rb(20)]] * 3,
                                                state=v_cycle(P, 3, [[_rb(18)]] * 3, [[_jac(14)]] * 3))),
    # systems of PDEs: linear elasticity, two unknowns, 9-point coupling
    "elasticity2d_L6_mixed": ("2D_FD_LinearElasticity_fromL2", 4, 6, elasticity_mixed_cycle),
    # the problem's own solver settings as a candidate: V(2,1), collective red-black w = 1.25 (exa3:10-23)
    "elasticity2d_L6_v21_rb": ("2D_FD_LinearElasticity_fromL2", 3, 6,
                               lambda P: v_cycle(P, 3, [[_rb(23)] * 2] * 3, [[_rb(23)]] * 3)),
    "elasticity2d_L5_blocks": ("2D_FD_LinearElasticity_fromL2", 3, 5,
                               lambda P: v_cycle(P, 2, [[("blocks", 12, ((2, 2), (2, 2)))], [("decoupled", 14), _jac(12)]],
                                                 [[("blocks", 10, ((1, 2), (1, 2))), ("decoupled_rb", 18)], [_rb(20)]])),
    # collective block smoothers whose block shape differs per unknown — most of the BlockShape terminals the grammar
    # registers for a two-unknown problem (grammar/multigrid.py:388-407: itertools.product over the unknowns' shapes)
    "elasticity2d_L5_mixed_shapes": ("2D_FD_LinearElasticity_fromL2", 3, 5,
                                     lambda P: v_cycle(P, 2, [[("blocks", 12, ((2, 1), (1, 1))), _rb(20)], [("blocks", 14, ((2, 2), (2, 1)))]],
                                                       [[("blocks", 10, ((1, 1), (1, 2))), _rb(20)], [("blocks", 9, ((3, 1), (1, 3))), _rb(18)]])),
    # a system of THREE PDEs (round 6): collective / decoupled / block smoothers over three unknowns
    "threespecies2d_L5_mixed": ("2D_FD_ThreeSpecies", 3, 5,
                                lambda P: v_cycle(P, 2, [[("decoupled", 14), _rb(20)], [("blocks", 12, ((2, 1), (2, 1), (2, 1)))]],
                                                  [[_jac(12), ("decoupled_rb", 18)], [_rb(22)]])),
    "threespecies2d_L6_v21_rb": ("2D_FD_ThreeSpecies", 3, 6,
                                 lambda P: v_cycle(P, 3, [[_rb(20)] * 2] * 3, [[_rb(20)]] * 3)),
    "poisson3d_L4_v22_block": ("3D_FD_Poisson_fromL2", 2, 4,
                               lambda P: v_cycle(P, 2, [[_blk(10, (2, 1, 1))] * 2] * 2, [[_blk(10, (2, 1, 1))] * 2] * 2)),
    "poisson3d_L4_mixed": ("3D_FD_Poisson_fromL2", 2, 4,
                           lambda P: v_cycle(P, 2, [[_rb(23)], [_blk(8, (2, 2, 1)), _jac(12)]],
                                             [[_jac(12), _rb(23)], [_blk(8, (1, 2, 2))]], cgc_index=19)),
}


# ---- non-linear FAS cycles (grammar/multigrid.py with FAS=True) ---------------------------------------
def fas_smooth(P, depth, cycle, spec):
    """spec: ('picard', idx, 'single'|'rb') | ('newton', idx, 'single'|'rb', n_newton_steps)"""
    partitioning = P["red_black"] if spec[2] == "rb" else P["single"]
    if spec[0] == "picard":
        return P[f"jacobi_picard_{depth}"](spec[1], partitioning, cycle)
    return P[f"jacobi_newton_{depth}"](spec[1], partitioning, spec[3], cycle)


def fas_v_cycle(P, depth, pre, post, cgc_index=18):
    """FAS V-cycle over depth+1 levels: pre/post = per level (finest first) lists of smoother specs."""
    state = P["u_and_f"]
    cycle = P["residual_0"](state)
    for d in range(depth):
        for spec in pre[d]:
            state = fas_smooth(P, d, cycle, spec)
            cycle = P[f"residual_{d}"](state)
        if d < depth - 1:
            cycle = P[f"coarsening_{d}"](P[f"A_{d + 1}"], P[f"zero_{d + 1}"], P[f"R_{d}"], cycle)
        else:
            state = P[f"correct_with_coarse_grid_solver_{d}"](cgc_index, P[f"P_{d + 1}"], P[f"CGS_{d + 1}"],
                                                              P[f"R_{d}"], cycle)
    for d in range(depth - 1, -1, -1):
        if d < depth - 1:
            state = P[f"update_with_coarse_grid_correction_{d}"](cgc_index, P[f"P_{d + 1}"], state, P[f"R_{d}"])
        for spec in post[d]:
            cycle = P[f"residual_{d}"](state)
            state = fas_smooth(P, d, cycle, spec)
    return state


def _nw(i, p="single", n=1): return ("newton", i, p, n)
def _pc(i, p="single"): return ("picard", i, p)


# name -> (depth, builder); levels come from the problem's .knowledge file (tests/golden/problems/FAS_2D_Exponential: 3..6)
FAS_CASES = {
    "fas2d_v11_newton": (3, lambda P: fas_v_cycle(P, 3, [[_nw(14)], [_nw(14)], [_nw(14)]], [[_nw(14)], [_nw(14)], [_nw(14)]])),
    "fas2d_v21_mixed": (3, lambda P: fas_v_cycle(P, 3, [[_nw(14), _pc(12, "rb")], [_pc(14, "rb")], [_nw(16, "rb", 2)]],
                                                 [[_nw(14, "rb")], [_pc(10)], [_nw(14, "single", 3)]], cgc_index=16)),
    "fas2d_two_grid_picard": (1, lambda P: fas_v_cycle(P, 1, [[_pc(14, "rb"), _pc(14, "rb")]], [[_pc(14, "rb")]])),
}
