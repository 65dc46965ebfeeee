"""pdhg_take_step_adaptive (the adaptive take_step with its scalar part in C) and the
one-graph-launch trial against the statement-by-statement Python loop over
pdhg_trial_step / pdhg_accept with separate launches: the four combinations must give
bitwise identical trajectories (same kernels, same reduction order, same scalar
arithmetic), and all of them must match the CPU oracle."""
import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import pagerank_lp, random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (AdaptiveStepsizeParams, PdhgSolverState,
                                                             take_step)
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _trajectory(p, steps, monkeypatch, graph, native):
    # graph: False = separate launches, True = one HIP-graph launch per trial, "one_kernel" = one persistent kernel
    monkeypatch.setenv("PDHG_GRAPH", "1" if graph else "0")
    monkeypatch.setenv("PDHG_COOP", "1" if graph == "one_kernel" else "0")
    monkeypatch.setenv("PDHG_PY_TAKE_STEP", "0" if native else "1")
    eng = HipPdhgEngine.from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    sizes = []
    for _ in range(steps):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
        sizes.append(st.step_size)
    x, y = eng.get_current()
    xa, ya = eng.get_average()
    out = (np.array(sizes), x, y, xa, ya, st.total_number_iterations, st.cumulative_kkt_passes)
    eng.close()
    return out


@pytest.mark.parametrize("maker", [lambda: random_lp(5000, 4000, 8, seed=7),
                                   lambda: pagerank_lp(20000, seed=2),
                                   lambda: H.skewed_lp(3000, 9000, seed=7, dense_rows=2, dense_cols=2),
                                   lambda: H.example_lp()],
                         ids=["random", "pagerank", "skewed_long_rows", "example_lp"])
def test_graph_and_native_take_step_are_bitwise_the_plain_path(gpu_required, monkeypatch, maker):
    p = maker()
    ref = _trajectory(p, 80, monkeypatch, graph=False, native=False)
    for graph, native in ((True, False), (False, True), (True, True), ("one_kernel", False), ("one_kernel", True)):
        got = _trajectory(p, 80, monkeypatch, graph=graph, native=native)
        if graph == "one_kernel":
            monkeypatch.setenv("PDHG_GRAPH", "1")
            monkeypatch.setenv("PDHG_COOP", "1")
            probe = HipPdhgEngine.from_problem(p)
            assert probe.layout_info()["trial_graph"] == 2        # the persistent-kernel path really ran
            probe.close()
        for a, b in zip(ref, got):
            assert np.array_equal(a, b), (graph, native)
    # and the oracle, over a short free-running stretch (reduction scalars agree to 1e-12
    # per step and long rows to 1e-13, so trajectories drift apart slowly: DESIGN.md section 2)
    short = _trajectory(p, 30, monkeypatch, graph="one_kernel", native=True)
    st = H.oracle_from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    st.step_size, st.primal_weight = step, pw
    for _ in range(30):
        st.take_step_adaptive(0.3, 0.6)
    assert st.total_number_iterations == short[5]
    np.testing.assert_allclose(short[1], st.x, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(short[2], st.y, rtol=1e-8, atol=1e-8)


def test_native_take_step_reports_zero_movement(gpu_required):
    """example_cc_lp reaches movement == 0 (test_primal_dual_hybrid_gradient.jl:391-412)."""
    p = H.example_cc_lp()
    eng = HipPdhgEngine.from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    for _ in range(200):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
        if st.numerical_error:
            break
    assert st.numerical_error

