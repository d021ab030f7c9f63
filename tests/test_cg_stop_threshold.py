"""The coarse CG's stopping rule is `sqrt(g) <= T` (T = tol * sqrt(g0); Poisson/2D_FD_Poisson_fromL2.exa3:12-14). Since round 6 the
device decides it on g itself: g <= g*, g* = the largest double whose correctly rounded square root is <= T, found once per
solve by stepping a few units in the last place around T*T (csrc/evomg_cg1.h: cg1_stop_threshold; falls back to the square root
when the search does not settle). This test restates that search in numpy (IEEE double, correctly rounded sqrt) and checks the
claim it rests on — the two rules decide alike for every g around g* — over wide ranges of T. The device function itself is
checked the same way on the GPU by tools/check_stop_threshold.hip (2^22 thresholds: no decision differs)."""
import numpy as np


def threshold(t):
    """(gstar, exact) as cg1_stop_threshold forms them"""
    t = np.float64(t)
    if not (t >= 0.0 and t < np.inf):
        return np.float64(0.0), False
    with np.errstate(over="ignore", under="ignore"):
        g = t * t
    if not g < np.inf:
        return np.float64(0.0), False
    up = lambda v: (v.view(np.int64) + 1).view(np.float64)
    down = lambda v: (v.view(np.int64) - 1).view(np.float64)
    g = np.array(g, dtype=np.float64)
    for _ in range(4):
        if g > 0.0 and np.sqrt(g) > t:
            g = down(g)
    for _ in range(4):
        u = up(g)
        if u < np.inf and np.sqrt(u) <= t:
            g = u
    u = up(g)
    exact = bool(np.sqrt(g) <= t and (not (u < np.inf) or np.sqrt(u) > t))
    return np.float64(g), exact


def test_threshold_on_the_dot_product_decides_like_the_square_root():
    rng = np.random.default_rng(6)
    ts = np.ldexp(1.0 + rng.random(20000), rng.integers(-700, 700, 20000))
    ts = np.concatenate([ts, [0.0, 4.9e-324, 1e-310, 1e-200, 1.5e-154, 1.3e154, 1e200, 1.7e308, 1e-12, 1.0, 3.0, 1e-6]])
    exact = 0
    with np.errstate(over="ignore", under="ignore", invalid="ignore"):
        for t in ts:
            gstar, ok = threshold(t)
            if not ok:
                continue
            exact += 1
            bits = np.array(gstar).view(np.int64)
            for d in range(-8, 9):
                b = bits + d
                if b < 0 or b >= 0x7FF0000000000000:
                    continue
                g = np.array(b, dtype=np.int64).view(np.float64)
                assert (not (g > gstar)) == (not (np.sqrt(g) > t)), (t, g, gstar)
            for g in (np.float64(0.0), gstar * 0.5, gstar * 2.0, np.float64(np.nan)):
                assert (not (g > gstar)) == (not (np.sqrt(g) > t))
    assert exact > 0.7 * len(ts)                     # the fallback is for thresholds whose square leaves the double range


def test_thresholds_the_search_refuses():
    for t in (-1.0, np.inf, np.nan, 1e200):          # negative, infinite, not a number, square overflows
        assert threshold(t)[1] is False
    assert threshold(0.0) == (0.0, True)             # tol = 0: stops only at g = 0
