"""test/test_trust_region_utils.jl restated for the host implementation
(firstorderlp.jl_amd/trust_region_utils.py).  CPU only."""
import numpy as np
import pytest

from firstorderlp_jl_amd.iteration_stats_utils import HostOps, compute_dual_stats, corrected_dual_obj
from firstorderlp_jl_amd.trust_region_utils import (bound_optimal_objective,
                                                    solve_bound_constrained_trust_region,
                                                    weighted_norm)
from tests import helpers as H
from tests import trust_region_cases as C

INF = np.inf
A = lambda *v: np.array(v, dtype=float)


@pytest.mark.parametrize("approx", [True, False])
def test_unbounded(approx):                               # :18-56
    r = solve_bound_constrained_trust_region(A(0.0), A(-1.0), A(-INF), A(INF), A(1.0), 5.0, approx)
    assert r.value == -5.0 and np.array_equal(r.solution, [5.0])
    r = solve_bound_constrained_trust_region(A(0.0, 0.0), A(1.0, 1.0), A(-INF, -INF), A(INF, INF),
                                             A(2.0, 1.0), np.sqrt(6.0), approx)
    np.testing.assert_allclose(r.solution, [-1.0, -2.0], atol=1e-8)
    assert abs(r.value + 3.0) <= 1e-8


def test_bounded_cases():                                 # :58-210
    s = lambda c, g, lo, hi, w, rad: solve_bound_constrained_trust_region(c, g, lo, hi, w, rad, False)
    assert np.array_equal(s(A(0.0), A(-1.0), A(-INF), A(INF), A(1.0), 5.0).solution, [5.0])
    assert np.array_equal(s(A(0.0), A(-1.0), A(-INF), A(0.0), A(1.0), 5.0).solution, [0.0])
    assert np.array_equal(s(A(0.0), A(-1.0), A(-INF), A(2.0), A(1.0), 5.0).solution, [2.0])
    np.testing.assert_allclose(s(A(0.0, 0.0), A(-2.0, -1.0), A(-INF, -INF), A(3.0, INF), A(1.0, 1.0), 5.0).solution,
                               [3.0, 4.0], atol=1e-8)
    assert np.array_equal(s(A(0.0, 0.0), A(-1.0, 0.0), A(-INF, -INF), A(2.0, INF), A(1.0, 1.0), 5.0).solution,
                          [2.0, 0.0])
    w = A(16.0, 9.0)
    r = s(A(0.0, 0.0), A(-4.0, -3.0), A(-INF, -INF), A(INF, INF), w, np.sqrt(2.0))
    assert abs(weighted_norm(r.solution, w) - np.sqrt(2.0)) <= 1e-8
    np.testing.assert_allclose(r.solution, [0.25, 1.0 / 3.0], atol=1e-8)
    n = 100
    for m in (10.0, 50.0):
        rad = np.sqrt(sum(min(i, m) ** 2 for i in range(1, n + 1)))
        r = s(np.zeros(n), -np.ones(n), np.zeros(n), 1.0 * np.arange(1, n + 1), np.ones(n), rad)
        np.testing.assert_allclose(r.solution, [min(i, m) for i in range(1, n + 1)], atol=1e-8)


@pytest.mark.parametrize("case", C.CASES, ids=lambda c: c[0])
def test_bound_optimal_objective_host(case):             # :212-327
    name, maker, x, y, radius, norm, expected = case
    p = maker()
    x, y = np.array(x), np.array(y)
    r = bound_optimal_objective(p, x, y, np.ones(len(x)), np.ones(len(y)), radius, norm, HostOps(p))
    C.check(r, expected, 0.0 if name != "euclid" else 1e-14)
    if name in ("r2_max", "corrected"):
        assert r.lower_bound_value == corrected_dual_obj(p, compute_dual_stats(p, x, y))
    if name == "euclid":
        d2 = float(np.sum((r.primal_solution - x) ** 2) + np.sum((r.dual_solution - y) ** 2))
        assert abs(d2 - radius ** 2) <= 1e-12
