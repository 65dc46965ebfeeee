"""Several adaptive take_steps per launch (steps_kernel: the accept / reject decision and the step-size rule on the
device, csrc/trial_kernel.hpp) against one launch per trial: the same statements (adaptive_step_rule is ONE
host/device function, the powers of the iteration count come from the host's pow on both paths), so the
trajectories must be bitwise identical -- step sizes, iterates, averages, counters -- whatever the batch sizes."""
import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import pagerank_lp, random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_steps
from tests import helpers as H

pytestmark = pytest.mark.gpu

POLICY = AdaptiveStepsizeParams(0.3, 0.6)


def _run(p, batches, monkeypatch, device_loop, relaxed=False, step_scale=1.0):
    monkeypatch.setenv("PDHG_DEVICE_LOOP", "1" if device_loop else "0")
    monkeypatch.setenv("PDHG_SMALL_LP", "0")              # (small LPs would otherwise take their batches in the LDS kernel)
    monkeypatch.setenv("PDHG_ROW_ORDER", "relaxed" if relaxed else "strict")
    eng = HipPdhgEngine.from_problem(p)
    assert eng.layout_info()["trial_graph"] == 2          # the persistent-kernel path is the one in use
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(eng, step_size=step * step_scale, primal_weight=pw)
    sizes = []
    for k in batches:
        done = take_steps(POLICY, st, k)
        assert done == k or st.numerical_error
        sizes.append(st.step_size)
        if st.numerical_error:
            break
    x, y = eng.get_current()
    xa, ya = eng.get_average()
    info = eng.average_info()
    out = (np.array(sizes), x, y, xa, ya, np.array(info), st.total_number_iterations, st.cumulative_kkt_passes,
           st.numerical_error)
    eng.close()
    return out


@pytest.mark.parametrize("maker,batches", [
    (lambda: random_lp(5000, 4000, 8, seed=7), [1, 2, 3, 40, 64, 7, 200]),
    (lambda: H.skewed_lp(3000, 9000, seed=7, dense_rows=2, dense_cols=2), [64, 64, 5, 100]),
    (lambda: pagerank_lp(20000, seed=2), [10, 64, 64, 64]),
    (lambda: H.example_lp(), [3, 50, 50]),
    (lambda: H.example_cc_lp(), [200]),                  # reaches movement == 0 inside a batch
], ids=["random", "skewed_long_rows", "pagerank", "example_lp", "zero_movement"])
def test_device_loop_is_bitwise_the_per_trial_launches(gpu_required, monkeypatch, maker, batches):
    p = maker()
    ref = _run(p, batches, monkeypatch, device_loop=False)
    got = _run(p, batches, monkeypatch, device_loop=True)
    for k, (a, b) in enumerate(zip(ref, got)):
        assert np.array_equal(a, b), k


def test_device_loop_hands_an_unfinished_take_step_back(gpu_required, monkeypatch):
    """A table of 3 powers per launch: launches end inside take_steps all the time; the host finishes them."""
    p = random_lp(5000, 4000, 8, seed=7)
    ref = _run(p, [40, 40], monkeypatch, device_loop=False, step_scale=300.0)    # far too long a first step: rejections
    monkeypatch.setenv("PDHG_STEPS_TEST_TABLE", "3")
    got = _run(p, [40, 40], monkeypatch, device_loop=True, step_scale=300.0)
    for k, (a, b) in enumerate(zip(ref, got)):
        assert np.array_equal(a, b), k
    assert got[6] > 80


def test_device_loop_barrier_timeout_hands_over_to_the_graph_path(gpu_required, monkeypatch, capfd):
    """A multi-step launch whose workgroups are not all co-resident (a grid four times what the device holds) cannot pass
    its first barrier: every spin is bounded, the launch reports zero trials, the host repeats them on the graph path
    and the handle stays there -- same trajectory as a handle that never used a persistent kernel."""
    p = random_lp(3000, 2500, 6, seed=11)
    monkeypatch.setenv("PDHG_COOP", "0")
    ref = None
    monkeypatch.setenv("PDHG_SMALL_LP", "0")
    monkeypatch.setenv("PDHG_ROW_ORDER", "strict")
    outs = []
    for pretend in (None, "8192"):
        monkeypatch.setenv("PDHG_COOP", "0" if pretend is None else "1")
        monkeypatch.setenv("PDHG_DEVICE_LOOP", "0" if pretend is None else "1")
        if pretend:
            monkeypatch.setenv("PDHG_COOP_TEST_PRETEND_WGS", pretend)
        eng = HipPdhgEngine.from_problem(p)
        step, pw = H.initial_step_and_weight(p)
        st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
        if pretend:
            assert eng.layout_info()["trial_graph"] == 2
        assert take_steps(POLICY, st, 30) == 30
        if pretend:
            assert eng.layout_info()["trial_graph"] == 1
            assert "timed out" in capfd.readouterr().err
        outs.append(eng.get_current() + eng.get_average() + (np.array([st.step_size, st.total_number_iterations]),))
        eng.close()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)


def test_device_loop_matches_the_oracle_in_exact_sums_mode(gpu_required, monkeypatch):
    """600 free-running steps in batches of 64 against the CPU restatement with exactly rounded sums: bitwise."""
    p = random_lp(4000, 3000, 7, seed=5)
    got = _run(p, [64] * 9 + [24], monkeypatch, device_loop=True)
    st = H.oracle_from_problem(p)
    st.exact_sums = True
    step, pw = H.initial_step_and_weight(p)
    st.step_size, st.primal_weight = step, pw
    for _ in range(600):
        st.take_step_adaptive(0.3, 0.6)
    assert st.total_number_iterations == got[6]
    assert st.step_size == got[0][-1]
    assert np.array_equal(got[1], st.x) and np.array_equal(got[2], st.y)


def test_device_loop_in_relaxed_order_and_through_optimize(gpu_required, monkeypatch):
    """Default row order; and a whole optimize() (evaluations, restarts between the batches) with and without the loop."""
    p = H.skewed_lp(3000, 9000, seed=3, dense_rows=1, dense_cols=3)
    ref = _run(p, [64, 64, 64], monkeypatch, device_loop=False, relaxed=True)
    got = _run(p, [64, 64, 64], monkeypatch, device_loop=True, relaxed=True)
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import PdhgParameters, optimize
    from firstorderlp_jl_amd.saddle_point import RestartScheme, RestartToCurrentMetric, construct_restart_parameters
    from firstorderlp_jl_amd.termination import construct_termination_criteria
    q = random_lp(3000, 2500, 6, seed=11)
    tc = construct_termination_criteria(eps_optimal_absolute=1e-6, eps_optimal_relative=1e-6, iteration_limit=3000)
    rp = construct_restart_parameters(RestartScheme.ADAPTIVE_NORMALIZED, RestartToCurrentMetric.GAP_OVER_DISTANCE_SQUARED,
                                      1000, 0.5, 0.1, 0.9, 0.5, False)
    params = PdhgParameters(10, False, 1.0, 1.0, True, 0, True, 40, tc, rp, AdaptiveStepsizeParams(0.3, 0.6))
    outs = []
    monkeypatch.setenv("PDHG_SMALL_LP", "0")
    for loop in ("0", "1"):
        monkeypatch.setenv("PDHG_DEVICE_LOOP", loop)
        o = optimize(params, q)
        outs.append((o.iteration_count, o.termination_reason, o.primal_solution, o.dual_solution))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    assert np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][3], outs[1][3])


# ---- round 4: the multi-step kernel's XCD-local mode (<= 32 items per product: every working workgroup on one XCD) ----
def _local_run(p, monkeypatch, local, bad=False, steps=(9, 40, 40)):
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import take_steps
    monkeypatch.setenv("PDHG_COOP_LOCAL", local)
    if bad:
        monkeypatch.setenv("PDHG_COOP_LOCAL_TEST_BAD", "1")
    eng = HipPdhgEngine.from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    sizes = []
    for k in steps:
        assert take_steps(AdaptiveStepsizeParams(0.3, 0.6), st, k) == k
        sizes.append(st.step_size)
    out = (np.array(sizes), st.total_number_iterations) + tuple(eng.get_current()) + tuple(eng.get_average())
    info = eng.layout_info()
    eng.close()
    if bad:
        monkeypatch.delenv("PDHG_COOP_LOCAL_TEST_BAD")
    return out, info


@pytest.mark.parametrize("shape", [(3000, 2500, 6), (6000, 5000, 5), (1800, 4000, 12)])
def test_xcd_local_mode_is_bitwise_the_all_xcd_kernel(gpu_required, monkeypatch, shape):
    p = random_lp(*shape, seed=13)
    ref, i0 = _local_run(p, monkeypatch, "0")
    got, i1 = _local_run(p, monkeypatch, "1")
    assert i0["device_loop"] == 1 and i0["steps_local"] == 0 and i1["steps_local"] == 1
    for k, (a, b) in enumerate(zip(ref, got)):
        assert np.array_equal(a, b), k


def test_xcd_local_mode_falls_back_when_its_barrier_cannot_complete(gpu_required, monkeypatch):
    p = random_lp(3000, 2500, 6, seed=14)
    ref, _ = _local_run(p, monkeypatch, "0")
    got, info = _local_run(p, monkeypatch, "1", bad=True)       # first launch times out (~0.1 s), the trial is repeated
    assert info["steps_local"] == 0 and info["device_loop"] == 1   # ... by the all-XCD kernel, which the handle keeps
    for k, (a, b) in enumerate(zip(ref, got)):
        assert np.array_equal(a, b), k
