"""Lazy accept: pdhg_accept leaves K7 (sum_x += w x', sum_y += w y'; saddle_point.jl:258-271)
to the kernels of the next trial, which read x and y anyway; every other entry point
settles it first.  The running sums must be bitwise what the separate accept_kernel
(PDHG_LAZY_ACCEPT=0) produces -- same two roundings per element, same order of additions
-- through every launch path (plain, one-graph trial, tiled sweep, slab passes, long rows,
shard groups) and through every interleaving of accept with the other entry points."""
import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import pagerank_lp, random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (AdaptiveStepsizeParams, PdhgSolverState,
                                                             take_step)
from tests import helpers as H

pytestmark = pytest.mark.gpu

LAYOUTS = {
    "plain": {"PDHG_GRAPH": "0"},
    "graph": {"PDHG_GRAPH": "1", "PDHG_COOP": "0"},          # one HIP-graph launch per trial
    "one_kernel": {"PDHG_GRAPH": "1", "PDHG_COOP": "1"},     # one persistent kernel per trial (trial_kernel.hpp)
    "tiled": {"PDHG_SPMV": "tiled", "PDHG_TILE_COLS": "700"},
    "python_take_step": {"PDHG_PY_TAKE_STEP": "1", "PDHG_GRAPH": "0"},
}
MAKERS = {"random": lambda: random_lp(5000, 4001, 8, seed=7),          # odd n: the scalar tail of primal_kernel
          "long_rows": lambda: H.skewed_lp(3000, 9000, seed=7, dense_rows=2, dense_cols=2),
          "pagerank": lambda: pagerank_lp(20000, seed=2)}


def _run(p, env, monkeypatch, lazy, steps=60, **kw):
    for k in ("PDHG_GRAPH", "PDHG_COOP", "PDHG_SPMV", "PDHG_TILE_COLS", "PDHG_SLAB_MB", "PDHG_PY_TAKE_STEP"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("PDHG_LAZY_ACCEPT", "1" if lazy else "0")
    eng = HipPdhgEngine.from_problem(p, **kw)
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    snapshots = []
    for k in range(steps):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
        if k in (0, 1, 17, steps - 1):          # reading the average settles a pending update mid-run
            snapshots.append(np.concatenate(eng.get_average()))
    out = (st.total_number_iterations, st.step_size, np.concatenate(eng.get_current()), *snapshots)
    eng.close()
    return out


@pytest.mark.parametrize("layout", sorted(LAYOUTS))
@pytest.mark.parametrize("name", sorted(MAKERS))
def test_lazy_accept_is_bitwise_the_separate_accept(gpu_required, monkeypatch, name, layout):
    p = MAKERS[name]()
    eager = _run(p, LAYOUTS[layout], monkeypatch, lazy=False)
    lazy = _run(p, LAYOUTS[layout], monkeypatch, lazy=True)
    assert eager[0] == lazy[0] and eager[1] == lazy[1]
    assert eager[0] > 60                        # some trials were rejected: a retried trial must not add twice
    for a, b in zip(eager[2:], lazy[2:]):
        assert np.array_equal(a, b)


def test_lazy_accept_with_column_slab_passes(gpu_required, monkeypatch):
    """The dual epilogue rides on the LAST slab pass only."""
    p = random_lp(150000, 140001, 8, seed=13)
    env = {"PDHG_SLAB_MB": "0.5", "PDHG_GRAPH": "1"}
    monkeypatch.setenv("PDHG_SLAB_MB", "0.5")
    eng = HipPdhgEngine.from_problem(p)
    assert eng.layout_info()["A_slabs"] >= 2 and eng.layout_info()["At_slabs"] >= 2
    eng.close()
    eager = _run(p, env, monkeypatch, lazy=False, steps=30)
    lazy = _run(p, env, monkeypatch, lazy=True, steps=30)
    for a, b in zip(eager, lazy):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("shards", [2, 3])
def test_lazy_accept_on_shard_groups(gpu_required, monkeypatch, shards):
    p = random_lp(6000, 5003, 8, seed=11)
    for env in ({}, {"PDHG_DIST_OVERLAP": "1", "PDHG_DIST_ROUND_WGS": "2", "PDHG_SPMV": "tiled", "PDHG_TILE_COLS": "900"}):
        eager = _run(p, env, monkeypatch, lazy=False, device_ids=[0] * shards)
        lazy = _run(p, env, monkeypatch, lazy=True, device_ids=[0] * shards)
        for a, b in zip(eager, lazy):
            assert np.array_equal(a, b)
        monkeypatch.delenv("PDHG_DIST_OVERLAP", raising=False)
        monkeypatch.delenv("PDHG_DIST_ROUND_WGS", raising=False)


def _script(eng, p, ops):
    """Drive the raw entry points; returns every array the script reads."""
    step, pw = H.initial_step_and_weight(p)
    rng = np.random.default_rng(5)
    seen = []
    for op in ops:
        if op == "trial":
            eng.trial_step(step, pw)
        elif op == "primal":
            eng.trial_primal(step, pw)
        elif op == "dual":
            eng.trial_dual(step, pw, 1.0)
        elif op == "accept":
            eng.accept(step)
            step *= 0.9
        elif op == "avg":
            seen.append(np.concatenate(eng.get_average()))
        elif op == "reset":
            eng.reset_average()
        elif op == "restart":
            eng.restart_to_average()
        elif op == "set":
            eng.set_current(rng.standard_normal(eng.n), np.abs(rng.standard_normal(eng.m)))
        elif op == "add_primal":
            eng.add_current_primal_to_average(0.25)
        elif op == "info":
            seen.append(np.array(eng.average_info(), dtype=float))
        seen.append(np.concatenate(eng.get_current())) if op in ("restart", "set") else None
    seen.append(np.concatenate(eng.get_current()))
    if eng.average_info()[0] > 0 and eng.average_info()[1] > 0:
        seen.append(np.concatenate(eng.get_average()))
    return seen


SCRIPTS = {
    "accept_twice": ["trial", "accept", "accept", "trial", "accept", "avg"],
    "accept_then_overwrite": ["trial", "accept", "set", "trial", "accept", "avg"],
    "accept_then_reset": ["trial", "accept", "trial", "accept", "reset", "trial", "accept", "trial", "accept", "avg"],
    "accept_then_restart": ["trial", "accept", "trial", "accept", "restart", "reset", "trial", "accept", "avg"],
    "split_trial": ["trial", "accept", "primal", "dual", "dual", "accept", "primal", "avg", "dual", "accept", "avg"],
    "rejections": ["trial", "accept", "trial", "trial", "trial", "accept", "trial", "avg", "trial", "accept", "avg"],
    "primal_only_weights": ["trial", "accept", "add_primal", "info", "trial", "accept", "add_primal", "info"],
}


@pytest.mark.parametrize("graph", ["0", "1", "one_kernel"])
@pytest.mark.parametrize("name", sorted(SCRIPTS))
def test_lazy_accept_interleavings(gpu_required, monkeypatch, name, graph):
    p = random_lp(700, 901, 6, seed=3)
    monkeypatch.setenv("PDHG_GRAPH", "1" if graph == "one_kernel" else graph)
    monkeypatch.setenv("PDHG_COOP", "1" if graph == "one_kernel" else "0")
    results = []
    for lazy in ("0", "1"):
        monkeypatch.setenv("PDHG_LAZY_ACCEPT", lazy)
        eng = HipPdhgEngine.from_problem(p)
        results.append(_script(eng, p, SCRIPTS[name]))
        eng.close()
    assert len(results[0]) == len(results[1]) > 0
    for a, b in zip(*results):
        assert np.array_equal(a, b)


def test_lazy_accept_device_evaluation_sees_the_settled_average(gpu_required, monkeypatch):
    """eval_point / distance_to_restart / trust_region_bound on the AVERAGE right after an accept."""
    from firstorderlp_jl_amd.evaluation import POINT_AVERAGE
    p = random_lp(900, 1100, 6, seed=9)
    m, n = p.constraint_matrix.shape
    outs = []
    for lazy in ("0", "1"):
        monkeypatch.setenv("PDHG_LAZY_ACCEPT", lazy)
        eng = HipPdhgEngine.from_problem(p)
        eng.set_original_problem(np.ones(m), np.ones(n), p.objective_vector, p.right_hand_side,
                                 p.variable_lower_bound, p.variable_upper_bound)
        step, pw = H.initial_step_and_weight(p)
        eng.save_restart_point()
        for _ in range(5):
            eng.trial_step(step, pw)
            eng.accept(step)
        outs.append((np.array(eng.eval_point(POINT_AVERAGE)), np.array(eng.distance_to_restart(POINT_AVERAGE)),
                     np.array(eng.trust_region_bound(POINT_AVERAGE, 1.0, 1.0, 0.5, 0)[:6])))
        eng.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
