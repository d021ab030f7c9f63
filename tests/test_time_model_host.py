"""Host logic of the contention-free time objective (ProgramGenerator.isolated_time_ms): which of the two C-ABI fits is asked
for, and when the cost of one coarse-solver iteration counts as known. No GPU: the handle is a stand-in that records the calls."""
import pytest

from evostencils_amd import problems
from evostencils_amd.code_generation.hip import ProgramGenerator


class _Handle:
    def __init__(self, per_coarse_values):
        self.per_coarse_values = list(per_coarse_values)
        self.calls = []

    def time_model(self, ops, n_short, n_long, per_coarse=None):
        self.calls.append((n_short, n_long, per_coarse))
        if per_coarse is not None:
            return 0.5, 0.2, per_coarse
        return 0.5, 0.2, self.per_coarse_values.pop(0)

    def time_iterations(self, ops, n):
        self.calls.append(("plain", n))
        return 0.3, n


def _generator(handle):
    g = ProgramGenerator.__new__(ProgramGenerator)           # no library, no device: only the fields the method reads
    g._problem = problems.poisson_2d(7, 11)
    g._coarse_iteration_ms = {}
    g._ensure_handle = lambda lo, hi, mapping: handle
    return g


def test_three_full_models_then_the_coarse_cost_is_known():
    h = _Handle([2.0e-3, 9.0e-3, 2.2e-3, 123.0])             # the median of the first three settles: 2.2 us
    g = _generator(h)
    times = [g.isolated_time_ms([], 7, 11, 100.0, coarse_iterations=40000) for _ in range(5)]
    assert [c[2] for c in h.calls] == [None, None, None, 2.2e-3, 2.2e-3]
    assert all(c[:2] == (2, 6) for c in h.calls)
    assert g.coarse_iteration_cost_ms(7, 11) == 2.2e-3
    assert times[3] == pytest.approx(0.5 + 0.2 * 100 + 2.2e-3 * 40000)
    assert times[1] == pytest.approx(0.5 + 0.2 * 100 + 9.0e-3 * 40000)      # before settling: the candidate's own fit


def test_fits_without_a_coarse_cost_do_not_settle_it_early():
    h = _Handle([0.0, 0.0, 2.0e-3, 0.0, 2.4e-3, 2.1e-3, 7.0])   # zero: a probe that blew up / a coarse solve that did not iterate
    g = _generator(h)
    for _ in range(7):
        g.isolated_time_ms([], 7, 11, 50.0, coarse_iterations=1000)
    assert [c[2] for c in h.calls] == [None] * 6 + [2.1e-3]


def test_a_coarse_solver_that_never_iterates_settles_on_zero():
    h = _Handle([0.0] * 12 + [5.0])
    g = _generator(h)
    for _ in range(13):
        g.isolated_time_ms([], 7, 11, 50.0, coarse_iterations=0)
    assert h.calls[-1][2] == 0.0 and g.coarse_iteration_cost_ms(7, 11) == 0.0


def test_the_coarse_cost_is_measured_again_every_so_often():
    """After COARSE_COST_REFRESH two-probe models the next candidate runs the full model again; the settled cost becomes the
    median of the last five positive samples (the stale first ones drop out one by one)."""
    h = _Handle([2.0e-3, 2.0e-3, 2.0e-3, 3.0e-3, 3.0e-3, 3.0e-3])
    g = _generator(h)
    n = 3 + 3 * (g.COARSE_COST_REFRESH + 1)
    for _ in range(n):
        g.isolated_time_ms([], 7, 11, 100.0, coarse_iterations=1000)
    full = [i for i, c in enumerate(h.calls) if c[2] is None]
    assert full == [0, 1, 2, 3 + g.COARSE_COST_REFRESH, 4 + 2 * g.COARSE_COST_REFRESH, 5 + 3 * g.COARSE_COST_REFRESH]
    assert g.coarse_iteration_cost_ms(7, 11) == 3.0e-3          # samples 2, 2, 3, 3, 3 of the last five
    assert h.calls[3][2] == 2.0e-3 and h.calls[-1][2] is None


def test_short_solves_and_missing_counts_fall_back_to_the_plain_probe():
    h = _Handle([])
    g = _generator(h)
    assert g.isolated_time_ms([], 7, 11, 2.0, coarse_iterations=10) == pytest.approx(0.3 * 2.0)
    assert g.isolated_time_ms([], 7, 11, 30.0, coarse_iterations=None) == pytest.approx(0.3 * 30.0)
    assert h.calls == [("plain", 2), ("plain", 6)]
