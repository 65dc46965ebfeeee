"""The reference's bound_optimal_objective known answers
(test/test_trust_region_utils.jl:212-327) through the DEVICE implementation
(pdhg_trust_region_bound)."""
import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine, _lib
from firstorderlp_jl_amd.evaluation import DeviceEvaluator, POINT_CURRENT
from firstorderlp_jl_amd.quadratic_programming import ScaledQpProblem
from firstorderlp_jl_amd.termination import cached_quadratic_program_info
from tests import trust_region_cases as C

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("small_eval,coop", [("1", "1"), ("0", "1"), ("0", "0")],
                         ids=["one_workgroup_kernel", "one_persistent_launch", "pass_by_pass"])
@pytest.mark.parametrize("case", C.CASES, ids=lambda c: c[0])
def test_bound_optimal_objective_device(gpu_required, monkeypatch, case, small_eval, coop):
    monkeypatch.setenv("PDHG_SMALL_EVAL", small_eval)
    monkeypatch.setenv("PDHG_TR_COOP", coop)
    name, maker, x, y, radius, norm, expected = case
    p = maker()
    eng = HipPdhgEngine.from_problem(p)
    sp_ = ScaledQpProblem(p, p, np.ones(p.num_constraints), np.ones(p.num_variables))
    ev = DeviceEvaluator(eng, sp_, cached_quadratic_program_info(p))
    eng.set_current(np.array(x), np.array(y))
    r = ev.bound(POINT_CURRENT, 1.0, 1.0, radius, norm, False)
    C.check(r, expected, 1e-12)
    assert (eng.layout_info()["tr_coop_calls"] > 0) == (small_eval == "0" and coop == "1")


@pytest.mark.parametrize("case", __import__("tests.stats_cases", fromlist=["CASES"]).CASES, ids=lambda c: c[0])
def test_convergence_information_device(gpu_required, case):
    """test/test_iteration_stats.jl:118-308: the ConvergenceInformation part through
    pdhg_eval_point (the device path evaluates the rays at the iterate itself, as
    optimize does, so the InfeasibilityInformation of those cases is host-only)."""
    from tests import stats_cases as S
    name, lp, x, y, xr, yr, want_ci, want_ii = case
    eng = HipPdhgEngine.from_problem(lp)
    sp_ = ScaledQpProblem(lp, lp, np.ones(lp.num_constraints), np.ones(lp.num_variables))
    ev = DeviceEvaluator(eng, sp_, cached_quadratic_program_info(lp))
    eng.set_current(np.array(x), np.array(y))
    from firstorderlp_jl_amd.termination import construct_termination_criteria
    from firstorderlp_jl_amd.solve_log import PointType
    tc = construct_termination_criteria(eps_optimal_absolute=1e-6, eps_optimal_relative=1e-6)
    st = ev.iteration_stats(POINT_CURRENT, tc, True, 6, 5.0, 1.5, 1.0, 1.0, PointType.POINT_TYPE_CURRENT_ITERATE)
    S.check_ci(st.convergence_information[0], want_ci, 1e-14)
    assert st.iteration_number == 5


# ---- round 4: the search as ONE persistent launch (csrc/tr_coop_kernel.hpp) against the pass-by-pass form ------------
def _bounds(p, monkeypatch, coop, steps=30, env=()):
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_steps
    from tests import helpers as H
    monkeypatch.setenv("PDHG_TR_COOP", coop)
    for k, v in env:
        monkeypatch.setenv(k, v)
    eng = HipPdhgEngine.from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    take_steps(AdaptiveStepsizeParams(0.3, 0.6), st, steps)
    eng.save_restart_point()
    take_steps(AdaptiveStepsizeParams(0.3, 0.6), st, steps)
    out = []
    for point in (0, 1):                    # current iterate, average
        for rng in (0, 1, 2):
            for radius in (0.0, 0.3, 5.0, 1e6):
                for approx in (0, 1):
                    out.append(np.array(eng.trust_region_bound(point, 1.7, 0.6, radius, rng, approx)))
    info = eng.layout_info()
    eng.close()
    for k, _ in env:
        monkeypatch.delenv(k)
    return np.array(out), info


@pytest.mark.parametrize("name", ["random", "pagerank", "qp"])
def test_one_launch_search_matches_pass_by_pass(gpu_required, monkeypatch, name):
    from firstorderlp_jl_amd.generators import pagerank_lp, random_lp
    from tests import helpers as H
    if name == "random":
        p = random_lp(40000, 30000, 6, seed=5)
    elif name == "pagerank":
        p = pagerank_lp(50000, seed=3)
    else:
        import scipy.sparse as sp
        from dataclasses import replace
        p = random_lp(9000, 12000, 5, seed=8)
        q = H.sparse_uniform(12000, 12000, 2e-4, 1)
        p = replace(p, objective_matrix=(q @ q.T + sp.identity(12000) * 0.1).tocsc())
    want, i0 = _bounds(p, monkeypatch, "0")
    got, i1 = _bounds(p, monkeypatch, "1")
    assert i0["tr_coop_calls"] == 0 and i1["tr_coop_calls"] == len(got)
    # [0] Lagrangian value, [1] / [2] primal / dual value change, [3] / [4] norms, [5] t*: sums grouped by another grid
    scale = np.maximum(np.abs(want).max(axis=0), 1e-300)
    np.testing.assert_allclose(got[:, :6] / scale[:6], want[:, :6] / scale[:6], rtol=0, atol=2e-12)
    assert np.array_equal(got[:, 6], want[:, 6])        # probe passes: the same search


def test_one_launch_search_falls_back_when_a_barrier_cannot_complete(gpu_required, monkeypatch):
    from firstorderlp_jl_amd.generators import random_lp
    p = random_lp(20000, 15000, 6, seed=6)
    want, _ = _bounds(p, monkeypatch, "0", steps=10)
    got, info = _bounds(p, monkeypatch, "1", steps=10, env=(("PDHG_TR_COOP_TEST_BAD_CENSUS", "1"),))
    assert info["tr_coop_calls"] == 0                    # the first call timed out (~0.1 s) and the handle left that form
    assert np.array_equal(got, want)                     # ... for the pass-by-pass form: the same bits


# ---- round 5: several searches in ONE persistent launch (pdhg_trust_region_bounds, tr_coop_batch_kernel) ----------------
def _state_with_restart_point(p, steps=30):
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_steps
    from tests import helpers as H
    eng = HipPdhgEngine.from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    take_steps(AdaptiveStepsizeParams(0.3, 0.6), st, steps)
    eng.save_restart_point()
    take_steps(AdaptiveStepsizeParams(0.3, 0.6), st, steps)
    return eng


@pytest.mark.parametrize("name", ["random", "pagerank", "qp", "small"])
def test_batched_searches_return_the_single_calls_bits(gpu_required, monkeypatch, name):
    """Three bounds of a restart check (average, current iterate, last restart point) and the two halves of MAX_NORM, asked
    for in one call: each row must be bit for bit what pdhg_trust_region_bound returns for that problem -- on medium single
    handles the batch shares one persistent launch (same statements, same grouping of the sums), elsewhere it is a loop."""
    from firstorderlp_jl_amd.generators import pagerank_lp, random_lp
    from tests import helpers as H
    if name == "random":
        p = random_lp(40000, 30000, 6, seed=5)
    elif name == "pagerank":
        p = pagerank_lp(50000, seed=3)
    elif name == "small":
        p = random_lp(900, 700, 5, seed=2)           # n + m <= 4 096: the one-workgroup kernel, call by call
    else:
        import scipy.sparse as sp
        from dataclasses import replace
        p = random_lp(9000, 12000, 5, seed=8)
        q = H.sparse_uniform(12000, 12000, 2e-4, 1)
        p = replace(p, objective_matrix=(q @ q.T + sp.identity(12000) * 0.1).tocsc())
    eng = _state_with_restart_point(p)
    requests = [([1, 0, 2], [0.3, 5.0, 0.7], [0, 0, 0]),       # average, current, restart point: the restart check
                ([0, 0], [1.0, 1.0], [1, 2]),                 # the halves of MAX_NORM at one point
                ([1], [2.5], [0]), ([2, 1, 0], [0.0, 1e6, 0.3], [0, 2, 1])]
    for approx in (0, 1):
        for points, radii, ranges in requests:
            got = eng.trust_region_bounds(points, 1.7, 0.6, radii, ranges, approx)
            for row, pt, rad, rg in zip(got, points, radii, ranges):
                want = np.array(eng.trust_region_bound(pt, 1.7, 0.6, rad, rg, approx))
                assert np.array_equal(row, want), (name, points, radii, ranges, approx)
    info = eng.layout_info()
    assert (info["tr_coop_calls"] > 0) == (name != "small")
    eng.close()


def test_batched_searches_fall_back_when_a_barrier_cannot_complete(gpu_required, monkeypatch):
    from firstorderlp_jl_amd.generators import random_lp
    p = random_lp(20000, 15000, 6, seed=6)
    monkeypatch.setenv("PDHG_TR_COOP", "0")
    eng = _state_with_restart_point(p, steps=10)
    want = eng.trust_region_bounds([1, 0, 2], 1.7, 0.6, [0.3, 5.0, 0.7], [0, 0, 0], 0)       # pass by pass, call by call
    eng.close()
    monkeypatch.setenv("PDHG_TR_COOP", "1")
    monkeypatch.setenv("PDHG_TR_COOP_TEST_BAD_CENSUS", "1")
    eng = _state_with_restart_point(p, steps=10)
    got = eng.trust_region_bounds([1, 0, 2], 1.7, 0.6, [0.3, 5.0, 0.7], [0, 0, 0], 0)
    assert eng.layout_info()["tr_coop_calls"] == 0       # the launch timed out (~0.1 s); the handle left the one-launch forms
    assert np.array_equal(got, want)
    eng.close()


def test_a_solve_takes_the_same_restart_decisions_with_and_without_the_batch(gpu_required, monkeypatch):
    """optimize() with solve_qp.jl's defaults (adaptive-normalized restarts: three bounds per check): iteration count,
    restart record and final objectives must not depend on whether the three searches share a launch."""
    from firstorderlp_jl_amd.generators import random_lp
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import optimize
    from tests.test_gpu_end_to_end import _params
    p = random_lp(12000, 10000, 8, seed=42)
    runs = {}
    for batch in ("1", "0"):
        monkeypatch.setenv("PDHG_TR_BATCH", batch)
        out = optimize(_params(1e-6, 40000), p)
        runs[batch] = (out.termination_string, out.iteration_count,
                       [str(s.restart_used) for s in out.iteration_stats],
                       out.iteration_stats[-1].convergence_information[0].primal_objective,
                       out.iteration_stats[-1].convergence_information[0].dual_objective)
    assert runs["1"] == runs["0"], (runs["1"][:2], runs["0"][:2])
    assert runs["1"][0] == "OPTIMAL"


@pytest.mark.own_row_order
def test_shared_wave_sums_have_the_bits_of_one_tree_per_quantity(gpu_required):
    """The check kernels' block reduction (csrc/eval_kernels.hpp: WaveSplit, round 6) against the form of rounds 1-5 on
    pseudo-random data of 80 binades, both signs and signed zeros: 8 / 16 / 22 / 30 / 64 quantities per lane, every wave
    total bit for bit (pdhg_selftest_wave_sums)."""
    from tests import helpers as H
    eng = HipPdhgEngine.from_problem(H.example_lp())
    for seed in (0, 1, 20260929):
        compared, differing = eng.selftest_wave_sums(seed)
        assert compared == 256 * 4 * (8 + 16 + 22 + 30 + 64) and differing == 0, (seed, compared, differing)
    eng.close()
