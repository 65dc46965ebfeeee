"""N1: the device-side evaluation branch against the host (numpy) evaluation on
the same engine state: convergence / infeasibility statistics, restart
distances and the trust-region objective bounds.  Tolerance 1e-9 relative
(sums are reduced in a different order; the trust-region breakpoint search is
a different algorithm with the same closed form at the end)."""
import dataclasses

import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.evaluation import (POINT_AVERAGE, POINT_CURRENT, POINT_RESTART,
                                            DeviceEvaluator, HostEvaluator)
from firstorderlp_jl_amd.generators import pagerank_lp, random_lp
from firstorderlp_jl_amd.preprocess import rescale_problem
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (
    AdaptiveStepsizeParams, EngineOps, PdhgSolverState, UnscaledEngineOps, take_step)
from firstorderlp_jl_amd.solve_log import PointType
from firstorderlp_jl_amd.termination import (cached_quadratic_program_info,
                                             construct_termination_criteria)
from firstorderlp_jl_amd.trust_region_utils import EUCLIDEAN_NORM, MAX_NORM
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _setup(p, ruiz=3, alpha=1.0, steps=35):
    sp_ = rescale_problem(ruiz, False, alpha, 0, p)
    eng = HipPdhgEngine.from_problem(sp_.scaled_qp)
    qp_cache = cached_quadratic_program_info(p)
    ev_h = HostEvaluator(eng, sp_, qp_cache, EngineOps(eng, sp_.scaled_qp), UnscaledEngineOps(eng, sp_))
    ev_d = DeviceEvaluator(eng, sp_, qp_cache)
    step, pw = H.initial_step_and_weight(sp_.scaled_qp)
    st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    for _ in range(steps):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
    return eng, ev_h, ev_d, st


def _close(a, b, rel=1e-9, scale=1.0):
    if np.isinf(a) or np.isinf(b):
        assert a == b
    else:
        assert abs(a - b) <= rel * max(abs(a), abs(b), scale), (a, b)


def _random_qp():
    import scipy.sparse as sp
    p = random_lp(2500, 3000, 7, 19)
    B = sp.random(3000, 3000, density=0.002, random_state=2, format="csc")
    p.objective_matrix = sp.csc_matrix(B.T @ B + sp.diags(np.random.default_rng(3).uniform(0.0, 2.0, 3000)))
    return p


@pytest.mark.parametrize("maker", [lambda: random_lp(3000, 4000, 8, 3),
                                   _random_qp,
                                   lambda: H.skewed_lp(2500, 6000, 5),
                                   lambda: pagerank_lp(20000, seed=2),
                                   lambda: H.example_lp()],
                         ids=["random", "random_qp", "skewed_freevars", "pagerank", "example_lp"])
def test_device_evaluation_matches_host(gpu_required, maker):
    eng, ev_h, ev_d, st = _setup(maker())
    tc = construct_termination_criteria()
    for point in (POINT_AVERAGE, POINT_CURRENT):
        a = ev_h.iteration_stats(point, tc, True, 36, 0.0, 70.0, st.step_size, st.primal_weight,
                                 PointType.POINT_TYPE_AVERAGE_ITERATE)
        b = ev_d.iteration_stats(point, tc, True, 36, 0.0, 70.0, st.step_size, st.primal_weight,
                                 PointType.POINT_TYPE_AVERAGE_ITERATE)
        ca, cb = a.convergence_information[0], b.convergence_information[0]
        obj_scale = abs(ca.primal_objective) + abs(ca.dual_objective) + 1.0
        for f in dataclasses.fields(ca):
            va, vb = getattr(ca, f.name), getattr(cb, f.name)
            if isinstance(va, float):
                _close(va, vb, scale=obj_scale if "objective" in f.name else 1e-6)
        ia, ib = a.infeasibility_information[0], b.infeasibility_information[0]
        for f in dataclasses.fields(ia):
            va, vb = getattr(ia, f.name), getattr(ib, f.name)
            if isinstance(va, float):
                _close(va, vb, scale=1e-6)
        assert a.iteration_number == b.iteration_number == 35

    wp = st.primal_weight / st.step_size
    wd = 1.0 / st.step_size / st.primal_weight
    for point in (POINT_AVERAGE, POINT_CURRENT, POINT_RESTART):
        dh, dd = ev_h.distance_sq_to_restart(point), ev_d.distance_sq_to_restart(point)
        _close(dh[0], dd[0]); _close(dh[1], dd[1])
        sh, sd = ev_h.point_sumsq(point), ev_d.point_sumsq(point)
        _close(sh[0], sd[0]); _close(sh[1], sd[1])
    dx2, dy2 = ev_h.distance_sq_to_restart(POINT_AVERAGE)
    radius = float(np.sqrt(wp * dx2 + wd * dy2))
    for point in (POINT_AVERAGE, POINT_CURRENT, POINT_RESTART):
        for norm in (EUCLIDEAN_NORM, MAX_NORM):
            for rad in (radius, 0.05 * radius, 30.0 * radius, 0.0):
                for approx in (False, True):
                    gh = ev_h.bound(point, wp, wd, rad, norm, approx)
                    gd = ev_d.bound(point, wp, wd, rad, norm, approx)
                    sc = abs(gh.lagrangian_value) + abs(gh.upper_bound_value - gh.lower_bound_value) + 1e-9
                    _close(gh.lagrangian_value, gd.lagrangian_value, scale=sc)
                    _close(gh.lower_bound_value, gd.lower_bound_value, rel=1e-8, scale=sc)
                    _close(gh.upper_bound_value, gd.upper_bound_value, rel=1e-8, scale=sc)


def test_device_restart_bookkeeping(gpu_required):
    eng, ev_h, ev_d, st = _setup(random_lp(2000, 1500, 6, 9), steps=20)
    xa, ya = eng.get_average()
    ev_d.restart(True)                       # current .= avg ; reset ; restart point .= current
    x, y = eng.get_current()
    assert np.array_equal(x, xa) and np.array_equal(y, ya)
    xr, yr = eng.get_point(POINT_RESTART)
    assert np.array_equal(xr, xa) and np.array_equal(yr, ya)
    assert eng.average_info()[:2] == (0, 0)
    assert ev_d.distance_sq_to_restart(POINT_CURRENT) == (0.0, 0.0)
    for _ in range(5):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
    dx2, dy2 = ev_d.distance_sq_to_restart(POINT_CURRENT)
    x, y = eng.get_current()
    assert abs(dx2 - float((x - xa) @ (x - xa))) <= 1e-12 * dx2
    assert abs(dy2 - float((y - ya) @ (y - ya))) <= 1e-12 * dy2


def test_cached_point_products_match_recomputation(gpu_required):
    """pdhg_eval_point / pdhg_trust_region_bound keep A*x and A'*y of the CURRENT
    and AVERAGE points until the state changes.  Engine `a` uses the cache the
    way optimize() does; engine `b` is forced to recompute before every call (a
    value-preserving set_current bumps the state version).  Every output must be
    bit-identical, across accepts, an average reset and a restart."""
    p = random_lp(20000, 16000, 8, 17)
    sp_ = rescale_problem(3, False, 1.0, 0, p)
    engines = []
    for _ in range(2):
        eng = HipPdhgEngine.from_problem(sp_.scaled_qp)
        DeviceEvaluator(eng, sp_, cached_quadratic_program_info(p))   # uploads the original problem
        step, pw = H.initial_step_and_weight(sp_.scaled_qp)
        engines.append((eng, PdhgSolverState(eng, step_size=step, primal_weight=pw)))
    (a, sa), (b, sb) = engines

    def probe(eng, invalidate):
        out = []
        for point in (POINT_AVERAGE, POINT_CURRENT, POINT_AVERAGE):
            for call in (lambda: eng.eval_point(point),
                         lambda: eng.trust_region_bound(point, 1.3, 0.7, 0.5, 0),   # both blocks
                         lambda: eng.trust_region_bound(point, 1.3, 0.7, 2.0, 1),   # primal half (MAX_NORM)
                         lambda: eng.trust_region_bound(point, 1.3, 0.7, 2.0, 2),   # dual half
                         lambda: np.array(eng.distance_to_restart(point))):
                if invalidate:
                    eng.set_current(None, None)
                out.append(call())
        return out

    def both_equal():
        for u, v in zip(probe(a, False), probe(b, True)):
            assert np.array_equal(u, v, equal_nan=True)

    for phase in range(4):
        for _ in range(7):
            take_step(AdaptiveStepsizeParams(0.3, 0.6), sa)
            take_step(AdaptiveStepsizeParams(0.3, 0.6), sb)
        both_equal()
        both_equal()          # a second round right away: everything served from the cache on `a`
        if phase == 1:
            a.save_restart_point(); b.save_restart_point()
            a.reset_average(); b.reset_average()
        if phase == 2:
            a.restart_to_average(); b.restart_to_average()
            a.reset_average(); b.reset_average()


@pytest.mark.parametrize("small_eval", ["1", "0"], ids=["one_workgroup_kernel", "pass_by_pass"])
def test_trust_region_search_branches_match_host(gpu_required, monkeypatch, small_eval):
    """Every exit of the breakpoint search -- no finite breakpoint at all, every finite breakpoint inside the ball
    (0 probe passes: the set-up pass's own sums), the bracket collapsing at t = 0, and the ordinary closed form --
    against the host implementation (the reference's median elimination restated in numpy), over six decades of radius.
    The value sums ride on the probes (sum g d min(t, thr)), the host clamps: same numbers to rounding."""
    import scipy.sparse as sp
    from firstorderlp_jl_amd import linear_programming_problem
    monkeypatch.setenv("PDHG_SMALL_EVAL", small_eval)     # n + m <= 4096: the whole search in one launch, or pass by pass
    rng = np.random.default_rng(5)
    base = random_lp(600, 800, 6, 4)
    n, m = base.num_variables, base.num_constraints
    free = linear_programming_problem(np.full(n, -np.inf), np.full(n, np.inf), base.objective_vector, 0.0,
                                      base.constraint_matrix, base.right_hand_side, m)       # all equalities, free variables
    boxed = linear_programming_problem(np.zeros(n), np.full(n, 0.5), base.objective_vector, 0.0,
                                       base.constraint_matrix, base.right_hand_side, 0)       # every direction meets a bound
    seen = set()
    for p in (free, boxed, base):
        eng, ev_h, ev_d, st = _setup(p, ruiz=0, alpha=None, steps=25)
        wp = st.primal_weight / st.step_size
        wd = 1.0 / st.step_size / st.primal_weight
        for point in (POINT_CURRENT, POINT_AVERAGE):
            for rad in (1e-9, 1e-4, 1e-2, 1.0, 1e2, 1e5):
                for rng_ in (0, 1, 2):
                    raw = eng.trust_region_bound(point, wp, wd, rad, rng_, False)
                    seen.add((p is free, int(raw[6]) == 0, raw[5] > 0.0))
                for norm in (EUCLIDEAN_NORM, MAX_NORM):
                    gh = ev_h.bound(point, wp, wd, rad, norm, False)
                    gd = ev_d.bound(point, wp, wd, rad, norm, False)
                    sc = abs(gh.lagrangian_value) + abs(gh.upper_bound_value - gh.lower_bound_value) + 1e-9
                    _close(gh.lower_bound_value, gd.lower_bound_value, rel=1e-9, scale=sc)
                    _close(gh.upper_bound_value, gd.upper_bound_value, rel=1e-9, scale=sc)
        eng.close()
    # both kinds of exit were taken: with and without probe passes
    assert any(zero_passes for _, zero_passes, _ in seen) and any(not zero_passes for _, zero_passes, _ in seen)


def test_eval_point_one_round_trip_is_bitwise_the_two_round_form(gpu_required, monkeypatch):
    """pdhg_eval_point on one handle reduces the row and the column statistics in ONE second stage (22 quantities side by
    side, one trip to the host); PDHG_EVAL_HOST_WORD=0 keeps the two rounds.  Same block partials, same order per
    quantity: not a bit may differ."""
    for p in (random_lp(30000, 20000, 6, seed=2), pagerank_lp(40000, seed=1)):
        eng, _, ev_d, _ = _setup(p)
        for point in (POINT_CURRENT, POINT_AVERAGE):
            one = np.array(eng.eval_point(point))
            monkeypatch.setenv("PDHG_EVAL_HOST_WORD", "0")
            two = np.array(eng.eval_point(point))
            monkeypatch.delenv("PDHG_EVAL_HOST_WORD")
            assert np.array_equal(one, two)
        eng.close()


def test_check_scalars_reduced_with_eval_point_are_bitwise_their_own_launches(gpu_required, monkeypatch):
    """pdhg_eval_point also reduces what the rest of a check asks for next -- the distances of the average and the current
    iterate to the last restart point (saddle_point.jl:432-477) and the sum of squares of the evaluated point
    (saddle_point.jl:1015-1047) -- and pdhg_distance_to_restart / pdhg_point_sumsq answer from that until the state moves
    (three of a check's six host round trips).  Same kernel, grid and per-quantity second stage as the calls' own launches:
    not a bit may differ (PDHG_EVAL_PREFETCH=0: the calls launch for themselves); a step, a new restart point or a
    different point must not be answered from stale values."""
    for p in (random_lp(30000, 20000, 6, seed=2), pagerank_lp(40000, seed=1)):
        eng, _, ev_d, st = _setup(p)
        eng.save_restart_point()
        for _ in range(7):
            take_step(AdaptiveStepsizeParams(0.3, 0.6), st)

        def check(point):
            ev = np.array(eng.eval_point(point))
            return (ev, eng.distance_to_restart(POINT_AVERAGE), eng.distance_to_restart(POINT_CURRENT), eng.point_sumsq(point),
                    eng.point_sumsq(POINT_CURRENT), eng.point_sumsq(POINT_RESTART))

        for point in (POINT_AVERAGE, POINT_CURRENT):
            fast = check(point)
            monkeypatch.setenv("PDHG_EVAL_PREFETCH", "0")
            slow = check(point)
            monkeypatch.delenv("PDHG_EVAL_PREFETCH")
            for a, b in zip(fast, slow):
                assert np.array_equal(np.asarray(a), np.asarray(b))
        # stale values: after a step / a new restart point the answers are those of fresh launches
        eng.eval_point(POINT_AVERAGE)
        take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
        after_step = (eng.distance_to_restart(POINT_AVERAGE), eng.distance_to_restart(POINT_CURRENT), eng.point_sumsq(POINT_AVERAGE))
        eng.eval_point(POINT_AVERAGE)
        eng.save_restart_point()
        after_save = (eng.distance_to_restart(POINT_AVERAGE), eng.distance_to_restart(POINT_CURRENT))
        assert tuple(after_save[1]) == (0.0, 0.0)
        monkeypatch.setenv("PDHG_EVAL_PREFETCH", "0")
        ref_save = (eng.distance_to_restart(POINT_AVERAGE), eng.distance_to_restart(POINT_CURRENT))
        monkeypatch.delenv("PDHG_EVAL_PREFETCH")
        assert np.array_equal(np.asarray(after_save), np.asarray(ref_save))
        x, y = eng.get_current()
        xa, ya = eng.get_average()
        xr, yr = eng.get_point(POINT_RESTART)
        np.testing.assert_allclose(np.asarray(after_step[2]), [xa @ xa, ya @ ya], rtol=1e-12)
        np.testing.assert_allclose(np.asarray(after_save[0]), [(xa - xr) @ (xa - xr), (ya - yr) @ (ya - yr)], rtol=1e-10, atol=1e-300)
        eng.close()
