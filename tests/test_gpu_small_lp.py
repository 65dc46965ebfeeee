"""Small LPs (n, m up to ~1400, rows of at most 256 entries): pdhg_take_steps_adaptive takes the whole batch in ONE
workgroup with every vector in LDS (csrc/small_lp_kernel.hpp).  The element arithmetic, the left-to-right row sums, the
double-double acceptance sums and the step rule are those of every other path, so the results must be bitwise the
per-trial launches' -- step sizes, iterates, averages, counters -- and the oracle's in exact-sums mode."""
import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_step, take_steps
from tests import helpers as H

pytestmark = pytest.mark.gpu

POLICY = AdaptiveStepsizeParams(0.3, 0.6)


def _run(p, batches, monkeypatch, small, relaxed=False, mix=False, step_scale=1.0, device_loop="0"):
    monkeypatch.setenv("PDHG_SMALL_LP", "1" if small else "0")
    # the reference runs: one launch per trial (small grids would default to the multi-step kernel)
    if device_loop is None:
        monkeypatch.delenv("PDHG_DEVICE_LOOP", raising=False)
    else:
        monkeypatch.setenv("PDHG_DEVICE_LOOP", device_loop)
    monkeypatch.setenv("PDHG_ROW_ORDER", "relaxed" if relaxed else "strict")
    eng = HipPdhgEngine.from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(eng, step_size=step * step_scale, primal_weight=pw)
    sizes = []
    for i, k in enumerate(batches):
        if mix and i % 2 == 1:
            for _ in range(k):                 # single steps between the batches: the lazy average update crosses the paths
                take_step(POLICY, st)
        else:
            done = take_steps(POLICY, st, k)
            assert done == k or st.numerical_error
        sizes.append(st.step_size)
        if st.numerical_error:
            break
    x, y = eng.get_current()
    xa, ya = eng.get_average()
    out = (np.array(sizes), x, y, xa, ya, np.array(eng.average_info()), st.total_number_iterations,
           st.cumulative_kkt_passes, st.numerical_error)
    eng.close()
    return out


@pytest.mark.parametrize("maker,batches", [
    (lambda: random_lp(30, 30, 3, seed=1), [2, 3, 64, 64, 300]),
    (lambda: random_lp(1200, 900, 6, seed=7), [64, 64, 7, 200]),
    (lambda: random_lp(700, 1400, 9, seed=3), [40, 40, 40]),
    (lambda: H.example_lp(), [3, 50, 50]),
    (lambda: H.example_cc_lp(), [200]),                   # movement == 0 inside a batch
    (lambda: H.skewed_lp(300, 900, seed=7, dense_rows=1, dense_cols=1, base_nnz=3), [64, 64]),   # rows of up to 256 entries at most? (else ineligible: still equal)
], ids=["tiny", "mid", "wide", "example_lp", "zero_movement", "skewed"])
def test_small_lp_batches_are_bitwise_the_per_trial_launches(gpu_required, monkeypatch, maker, batches):
    p = maker()
    ref = _run(p, batches, monkeypatch, small=False)
    got = _run(p, batches, monkeypatch, small=True)
    for k, (a, b) in enumerate(zip(ref, got)):
        assert np.array_equal(a, b), k
    mixed = _run(p, batches, monkeypatch, small=True, mix=True)
    for k, (a, b) in enumerate(zip(ref, mixed)):
        assert np.array_equal(a, b), ("mixed", k)
    relaxed_ref = _run(p, batches, monkeypatch, small=False, relaxed=True)
    relaxed_got = _run(p, batches, monkeypatch, small=True, relaxed=True)
    for k, (a, b) in enumerate(zip(relaxed_ref, relaxed_got)):
        assert np.array_equal(a, b), ("relaxed", k)


def test_a_launch_that_ends_inside_a_take_step_is_finished_by_the_host(gpu_required, monkeypatch):
    """The table of powers bounds the trials of a launch; when it runs out after a rejected trial the kernel hands the
    unfinished take_step back (its step size on entry rides in the result words) and the host finishes it launch by
    launch.  With a table of 3 entries that happens all the time: same trajectory."""
    p = random_lp(300, 250, 5, seed=2)
    ref = _run(p, [40, 40, 40], monkeypatch, small=False, step_scale=300.0)     # far too long a first step: rejections
    monkeypatch.setenv("PDHG_STEPS_TEST_TABLE", "3")
    got = _run(p, [40, 40, 40], monkeypatch, small=True, step_scale=300.0)
    for k, (a, b) in enumerate(zip(ref, got)):
        assert np.array_equal(a, b), k
    assert got[6] > 120                      # there were rejected trials, i.e. launches that ended inside a take_step


@pytest.mark.parametrize("device_loop", [None, "1"], ids=["default", "forced"])
def test_an_unfinished_take_step_never_reaches_the_multi_step_kernel(gpu_required, monkeypatch, device_loop):
    """The shipped configuration has BOTH batch kernels on (small-LP kernel first, multi-step kernel as the next
    choice).  A small-LP launch that ends inside a take_step must be finished launch by launch with its step size on
    entry as the average's weight (pdhg.jl:512) -- a fresh multi-step launch would start a new take_step and weigh the
    accept with the already reduced step size: the averages would silently drift from the reference."""
    p = random_lp(300, 250, 5, seed=2)
    ref = _run(p, [40, 40, 40], monkeypatch, small=False, step_scale=300.0)
    monkeypatch.setenv("PDHG_STEPS_TEST_TABLE", "3")
    got = _run(p, [40, 40, 40], monkeypatch, small=True, step_scale=300.0, device_loop=device_loop)
    for k, (a, b) in enumerate(zip(ref, got)):
        assert np.array_equal(a, b), k
    assert got[6] > 120


def test_small_lp_matches_the_oracle_in_exact_sums_mode(gpu_required, monkeypatch):
    p = random_lp(800, 600, 7, seed=5)
    got = _run(p, [64] * 9 + [24], monkeypatch, small=True)
    st = H.oracle_from_problem(p)
    st.exact_sums = True
    step, pw = H.initial_step_and_weight(p)
    st.step_size, st.primal_weight = step, pw
    for _ in range(600):
        st.take_step_adaptive(0.3, 0.6)
    assert st.total_number_iterations == got[6]
    assert st.step_size == got[0][-1]
    assert np.array_equal(got[1], st.x) and np.array_equal(got[2], st.y)


def test_small_lp_through_optimize(gpu_required, monkeypatch):
    """A whole optimize() -- rescaling, evaluations, restarts between the batches -- with and without the small-LP kernel."""
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import PdhgParameters, optimize
    from firstorderlp_jl_amd.saddle_point import RestartScheme, RestartToCurrentMetric, construct_restart_parameters
    from firstorderlp_jl_amd.termination import construct_termination_criteria
    q = random_lp(900, 700, 6, seed=11)
    tc = construct_termination_criteria(eps_optimal_absolute=1e-6, eps_optimal_relative=1e-6, iteration_limit=4000)
    rp = construct_restart_parameters(RestartScheme.ADAPTIVE_NORMALIZED, RestartToCurrentMetric.GAP_OVER_DISTANCE_SQUARED,
                                      1000, 0.5, 0.1, 0.9, 0.5, False)
    params = PdhgParameters(10, False, 1.0, 1.0, True, 0, True, 40, tc, rp, AdaptiveStepsizeParams(0.3, 0.6))
    outs = []
    monkeypatch.setenv("PDHG_DEVICE_LOOP", "0")
    for small in ("0", "1"):
        monkeypatch.setenv("PDHG_SMALL_LP", small)
        o = optimize(params, q)
        outs.append((o.iteration_count, o.termination_reason, o.primal_solution, o.dual_solution))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    assert np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][3], outs[1][3])
