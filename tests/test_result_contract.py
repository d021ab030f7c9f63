"""Result contract of the path: what `parse_output` + `evaluate` make of the generated
executable's stdout (exastencils.py:417-443,539-584) versus what our backend derives from the residual
history. Synthetic stdout in the format ExaStencils prints; where the reference checkout is present its own
`parse_output` is the judge, elsewhere a frozen table of its answers."""
import math
import os
import sys

import numpy as np
import pytest

from evostencils_amd import abi
from evostencils_amd.code_generation.hip import ProgramGenerator

INF = 1e100


def stdout_for(history, hit_cap, time_ms, max_its=100):
    lines = [f"Starting residual: {history[0]}"]
    for i in range(1, len(history)):
        lines.append(f"Residual after {i} iterations is {history[i]} --- convergence factor is {history[i] / history[i-1]}")
    if hit_cap:
        lines.append(f"Maximum number of solver iterations ( {max_its} ) was exceeded")
    lines.append(f"Total time to solve in {len(history) - 1} steps: {time_ms} ms")
    return "\n".join(lines)


def outcome(history, status, time_ms):
    h = np.asarray(history, dtype=float)
    return abi.SolveOutcome(time_ms, 0.0, float(len(h) - 1), h[0], h[-1], status, h)


CASES = {
    "converged": ([1.0, 0.1, 0.012, 0.0011, 1e-13], abi.SOLVE_CONVERGED, False),
    "iteration_cap": ([1.0] + [0.97 ** k for k in range(1, 101)], abi.SOLVE_MAXITER, True),
    "diverging": ([1.0, 3.0, 11.0, 40.0], abi.SOLVE_MAXITER, True),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_interpret_outcome_follows_parse_output(name):
    history, status, cap = CASES[name]
    t, rho, its = ProgramGenerator.interpret_outcome(outcome(history, status, 12.5), INF, solver_iteration_limit=500)
    n = len(history) - 1
    expected_rho = 1.0
    for a, b in zip(history[:-1], history[1:]):
        expected_rho *= math.pow(b / a, 1.0 / n)
    assert t == 12.5
    assert rho == pytest.approx(expected_rho, rel=1e-14)
    assert its == (INF if cap else n)
    ref = "/root/reference"
    if os.path.isdir(os.path.join(ref, "evostencils")):
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
        import generate_golden as gg
        gg.install_deap_stub()
        sys.path.insert(0, ref)
        try:
            from evostencils.code_generation.exastencils import ProgramGenerator as Ref
            rt, rrho, rits = Ref.parse_output(stdout_for(history, cap, 12.5), INF, 500)
        finally:
            sys.path.remove(ref)
            for m in [m for m in sys.modules if m.startswith("deap") or m.startswith("evostencils.") or m == "evostencils"]:
                del sys.modules[m]
        assert (rt, rits) == (t, its)
        assert rrho == pytest.approx(rho, rel=1e-12)


def test_nan_and_zero_iteration_sentinels():
    # NaN residual -> rho_inf = sqrt(infinity) enters the mean (exastencils.py:550-551) -> rho > 1 -> early return
    t, rho, its = ProgramGenerator.interpret_outcome(outcome([1.0, 0.5, float("nan")], abi.SOLVE_DIVERGED, 3.0), INF)
    assert its == INF and rho > 1
    # zero iterations -> infinity (exastencils.py:580-581)
    t, rho, its = ProgramGenerator.interpret_outcome(outcome([0.0], abi.SOLVE_CONVERGED, 0.1), INF)
    assert its == INF and rho == INF
    # iteration limit of the optimizer (scripts/optimize.py:32)
    t, rho, its = ProgramGenerator.interpret_outcome(outcome([1.0, 0.5, 0.25, 0.12], abi.SOLVE_CONVERGED, 1.0), INF, solver_iteration_limit=3)
    assert its == INF


def test_krylov_outer_convention():
    """Helmholtz prints ONE factor |res|/|initRes| and the 0-based step (exa3:160,193)."""
    hist = [4.0, 1.0, 0.2, 3e-7]
    t, rho, its = ProgramGenerator.interpret_outcome(outcome(hist, abi.SOLVE_CONVERGED, 9.0), INF, 10000, krylov_outer=True)
    assert rho == pytest.approx(3e-7 / 4.0) and its == 2
