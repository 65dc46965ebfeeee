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



@pytest.mark.parametrize("maker", [lambda: random_lp(5000, 4000, 8, seed=7),
                                   lambda: H.skewed_lp(3000, 9000, seed=7, dense_rows=2, dense_cols=2)],
                         ids=["random", "skewed_long_rows"])
def test_malitsky_pock_split_through_the_one_kernel_trial_is_bitwise_the_plain_path(gpu_required, monkeypatch, maker):
    """take_step(::MalitskyPockStepsizeParameters) (pdhg.jl:555-647): x' once (pdhg_trial_primal), then
    repeated dual trials (pdhg_trial_dual: xbar + K3..K6).  The dual trials run as ONE persistent
    kernel too (its phase 0 is xbar alone); results must be the bits of the separate launches."""
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import MalitskyPockStepsizeParameters
    p = maker()
    runs = {}
    for path in ("plain", "one_kernel"):
        monkeypatch.setenv("PDHG_GRAPH", "0" if path == "plain" else "1")
        monkeypatch.setenv("PDHG_COOP", "1" if path == "one_kernel" else "0")
        eng = HipPdhgEngine.from_problem(p)
        assert eng.layout_info()["trial_graph"] == (2 if path == "one_kernel" else 0)
        step, pw = H.initial_step_and_weight(p)
        st = PdhgSolverState(eng, step_size=step, primal_weight=pw, ratio_step_sizes=1.0)
        mp = MalitskyPockStepsizeParameters(downscaling_factor=0.7, breaking_factor=0.99, interpolation_coefficient=1.0)
        steps = []
        for _ in range(60):
            take_step(mp, st)
            steps.append(st.step_size)
        runs[path] = (np.array(steps), *eng.get_current(), *eng.get_average(), st.total_number_iterations)
        eng.close()
    for a, b in zip(runs["plain"], runs["one_kernel"]):
        assert np.array_equal(a, b)


def test_two_handles_driving_one_kernel_trials_from_two_threads(gpu_required, monkeypatch):
    """Two persistent trial kernels that are each only partly resident would wait for one another's
    workgroups; the library runs one at a time per device (a mutex from launch to results).  Two host
    threads, one handle each, must both finish with the bits of a handle run alone."""
    import threading
    monkeypatch.setenv("PDHG_GRAPH", "1")
    monkeypatch.setenv("PDHG_COOP", "1")
    p = random_lp(40000, 30000, 8, seed=9)

    def solo():
        eng = HipPdhgEngine.from_problem(p)
        assert eng.layout_info()["trial_graph"] == 2
        step, pw = H.initial_step_and_weight(p)
        st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
        for _ in range(150):
            take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
        out = (np.concatenate(eng.get_current()), st.total_number_iterations, st.step_size)
        eng.close()
        return out

    ref = solo()
    results, errors = [None, None], []

    def work(k):
        try:
            results[k] = solo()
        except Exception as exc:      # surfaced below
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    for r in results:
        assert r is not None and np.array_equal(r[0], ref[0]) and r[1:] == ref[1:]


def _random_qp(m=4000, n=3000, seed=4):
    """random_lp plus a sparse positive semidefinite objective matrix Q = B'B + diag."""
    import scipy.sparse as sp
    from firstorderlp_jl_amd.quadratic_programming import QuadraticProgrammingProblem
    p = random_lp(m, n, 7, seed=seed)
    B = sp.random(n // 3, n, density=3.0 / n, format="csr", random_state=seed + 1)
    Q = (B.T @ B + sp.diags(np.linspace(0.0, 0.5, n))).tocsc()
    Q.sort_indices()
    return QuadraticProgrammingProblem(
        variable_lower_bound=p.variable_lower_bound, variable_upper_bound=p.variable_upper_bound,
        objective_matrix=Q, objective_vector=p.objective_vector, objective_constant=0.0,
        constraint_matrix=p.constraint_matrix, right_hand_side=p.right_hand_side, num_equalities=p.num_equalities)


@pytest.mark.parametrize("policy", ["adaptive", "malitsky_pock"])
def test_qp_through_the_one_kernel_trial_is_bitwise_the_plain_path(gpu_required, monkeypatch, policy):
    """QPs (pdhg.jl:536-541, saddle_point.jl:1093-1100): the one-launch kernel gets a phase for Q x in front of
    the primal step and computes Q'dx and dx.(Q'dx) beside the two constraint-matrix products."""
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import MalitskyPockStepsizeParameters
    p = _random_qp()
    runs = {}
    for path in ("plain", "one_kernel"):
        monkeypatch.setenv("PDHG_GRAPH", "0" if path == "plain" else "1")
        monkeypatch.setenv("PDHG_COOP", "1" if path == "one_kernel" else "0")
        eng = HipPdhgEngine.from_problem(p)
        assert eng.layout_info()["trial_graph"] == (2 if path == "one_kernel" else 0)
        step, pw = H.initial_step_and_weight(p)
        if policy == "adaptive":
            st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
            pol = AdaptiveStepsizeParams(0.3, 0.6)
        else:
            st = PdhgSolverState(eng, step_size=step, primal_weight=pw, ratio_step_sizes=1.0)
            pol = MalitskyPockStepsizeParameters(downscaling_factor=0.7, breaking_factor=0.99, interpolation_coefficient=1.0)
        steps = []
        for _ in range(60):
            take_step(pol, st)
            steps.append(st.step_size)
        runs[path] = (np.array(steps), *eng.get_current(), *eng.get_average(), st.total_number_iterations)
        eng.close()
    assert runs["plain"][-1] >= 60
    for a, b in zip(runs["plain"], runs["one_kernel"]):
        assert np.array_equal(a, b)


def test_batched_take_steps_equal_single_calls(gpu_required):
    """pdhg_take_steps_adaptive(n) == n x pdhg_take_step_adaptive, bit for bit, including
    the early return on the step that reaches movement == 0."""
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import take_steps
    policy = AdaptiveStepsizeParams(0.3, 0.6)
    for p, n in ((random_lp(3000, 2500, 6, seed=3), 75), (H.example_cc_lp(), 200)):
        step, pw = H.initial_step_and_weight(p)
        one = HipPdhgEngine.from_problem(p)
        s1 = PdhgSolverState(one, step_size=step, primal_weight=pw)
        done1 = 0
        while done1 < n and not s1.numerical_error:
            take_step(policy, s1)
            done1 += 1
        many = HipPdhgEngine.from_problem(p)
        s2 = PdhgSolverState(many, step_size=step, primal_weight=pw)
        done2 = take_steps(policy, s2, n)
        assert done2 == done1
        assert s2.numerical_error == s1.numerical_error
        assert (s2.step_size, s2.total_number_iterations, s2.cumulative_kkt_passes) == \
            (s1.step_size, s1.total_number_iterations, s1.cumulative_kkt_passes)
        for a, b in zip(one.get_current() + one.get_average(), many.get_current() + many.get_average()):
            assert np.array_equal(a, b)
        one.close()
        many.close()
    eng = HipPdhgEngine.from_problem(H.example_lp())
    st = PdhgSolverState(eng, step_size=0.1, primal_weight=1.0)
    assert take_steps(policy, st, 0) == 0 and st.total_number_iterations == 0
    with pytest.raises(Exception):
        eng.take_steps_adaptive(-1, 0.3, 0.6, 0.1, 1.0, 0, 0.0)
    eng.close()


def test_barrier_timeout_falls_back_to_the_graph_path(gpu_required, monkeypatch, capfd):
    """A persistent trial kernel whose workgroups are not all co-resident (here: a grid four
    times what the device holds) cannot complete its grid barriers.  Every spin is bounded;
    the trial is repeated on the graph path and the handle stays there -- with the same
    results as a handle that never used the persistent kernel."""
    p = random_lp(3000, 2500, 6, seed=11)
    ref = _trajectory(p, 12, monkeypatch, graph=True, native=True)
    monkeypatch.setenv("PDHG_GRAPH", "1")
    monkeypatch.setenv("PDHG_COOP", "1")
    monkeypatch.setenv("PDHG_PY_TAKE_STEP", "0")
    monkeypatch.setenv("PDHG_COOP_TEST_PRETEND_WGS", "8192")
    eng = HipPdhgEngine.from_problem(p)
    assert eng.layout_info()["trial_graph"] == 2
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    sizes = []
    for _ in range(12):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
        sizes.append(st.step_size)
    assert eng.layout_info()["trial_graph"] == 1
    assert "timed out" in capfd.readouterr().err
    x, y = eng.get_current()
    xa, ya = eng.get_average()
    for a, b in zip(ref, (np.array(sizes), x, y, xa, ya, st.total_number_iterations, st.cumulative_kkt_passes)):
        assert np.array_equal(a, b)
    eng.close()


def test_barrier_error_mid_run_keeps_the_deferred_average(gpu_required):
    """The failing launch carries the previous step's deferred average update (lazy K7); the
    repeat on the graph path must not apply it twice nor lose it.  Run in a child process:
    the knob is read once per process."""
    import subprocess
    import sys
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import folp_loader
folp_loader.load()
from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_step
from tests import helpers as H
p = random_lp(3000, 2500, 6, seed=11)
def run(env):
    os.environ.update(env)
    eng = HipPdhgEngine.from_problem(p)
    kind = eng.layout_info()["trial_graph"]
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    for _ in range(30):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
    out = eng.get_current() + eng.get_average() + (np.array([st.step_size, st.total_number_iterations]),)
    after = eng.layout_info()["trial_graph"]
    eng.close()
    return kind, after, out
k0, a0, ref = run({"PDHG_COOP": "0"})
k1, a1, got = run({"PDHG_COOP": "1"})
assert (k0, a0, k1, a1) == (1, 1, 2, 1), (k0, a0, k1, a1)
for a, b in zip(ref, got):
    assert np.array_equal(a, b)
print("same")
'''
    import os
    env = dict(os.environ, PDHG_COOP_TEST_BREAK_AT="7", PDHG_ROW_ORDER="strict", PDHG_GRAPH="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "same" in r.stdout, r.stdout + r.stderr
    assert "timed out" in r.stderr
