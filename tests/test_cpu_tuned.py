"""The tuned CPU solver bench.py times beside the parity oracle (oracle/evomg_cpu_tuned.c: plain OpenMP, compiler-contracted
arithmetic, natural summation order) solves the headline workload's problem with the headline's algorithm: on a size the
oracle finishes in a blink it takes the same number of cycles as the oracle and its residual history agrees to rounding."""
import numpy as np

from evostencils_amd import abi, candidates, problems
from oracle import oracle


def test_tuned_cpu_solver_follows_the_oracle_to_rounding():
    lo, hi = 3, 7
    pd = problems.poisson_2d(lo, hi)
    ops = candidates.v_cycle_ops(lo, hi, 1, 1)
    h = oracle.create(pd, jacobi_semantics="textbook")
    ref = h.evaluate(ops, 1, 0, 128)
    want = h.get_field(hi, abi.F_SOL)
    u, hist, coarse_its, _seconds = oracle.tuned_poisson2d(lo, hi, pd.initial_solution(hi), pd.right_hand_side(hi), maxit=pd.max_iterations)
    assert len(hist) - 1 == int(ref.n_iterations) and ref.status == abi.SOLVE_CONVERGED
    assert hist[-1] <= 1e-12 * hist[0]
    assert np.abs(hist - ref.history).max() <= 1e-10 * ref.history[0]
    assert np.allclose(hist[:6], ref.history[:6], rtol=1e-9, atol=0)
    assert np.abs(u - want).max() <= 1e-10 * np.abs(want).max()
    assert coarse_its > 0
