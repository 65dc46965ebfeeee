"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.

Tolerances (fp64):
  * elementwise updates and SpMV rows of <= 2048 nonzeros: BIT-EXACT (the
    stream kernel adds each row's products in the oracle's order, no FMA);
  * rows split across workgroups (> 2048 nnz): |diff| <= 1e-13 * sum|a_ij x_j|;
  * the reduction scalars (wave/tree order): relative 1e-12;
  * K-step adaptive trajectories: relative 1e-9 on iterates (a 1-ulp scalar
    difference perturbs the step size, not the fixed point).
"""
import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (
    AdaptiveStepsizeParams, ConstantStepsizeParams,
    MalitskyPockStepsizeParameters, PdhgSolverState, take_step)
from oracle import oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _mk(p):
    return HipPdhgEngine.from_problem(p), H.oracle_from_problem(p)


def _spmv_abs_bound(A, x):
    return abs(A) @ np.abs(x)


@pytest.mark.parametrize("m,n,k,seed", [(1, 1, 1, 0), (7, 5, 3, 1),
                                        (300, 400, 10, 2), (5000, 3000, 10, 3),
                                        (20000, 20000, 10, 4),
                                        (1000, 50, 40, 5)])
def test_spmv_bit_exact_short_rows(gpu_required, m, n, k, seed):
    p = random_lp(m, n, min(k, n), seed)
    A = p.constraint_matrix
    eng = HipPdhgEngine.from_problem(p)
    info = eng.layout_info()
    assert info["A_long_rows"] == 0 and info["At_long_rows"] == 0
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n)
    y = rng.standard_normal(m)
    # (1000 x 50 with 40 per row: the transposed side has rows of ~800 entries -- bitwise in strict order, summed by
    #  their wave, within 1e-13 * sum |a x|, in the shipped relaxed order)
    H.assert_products_match_oracle(eng, A, x, y)


def test_spmv_long_rows_and_empty_rows(gpu_required):
    p = H.skewed_lp(3000, 9000, seed=7, dense_rows=2, dense_cols=2)
    A = p.constraint_matrix
    m, n = A.shape
    eng = HipPdhgEngine.from_problem(p)
    info = eng.layout_info()
    assert info["A_long_rows"] == 2 and info["At_long_rows"] == 2
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    ref = orc.spmv(m, n, A.indptr, A.indices, A.data, x)
    ref_t = orc.spmv_t(m, n, A.indptr, A.indices, A.data, y)
    got, got_t = eng.spmv(x), eng.spmv_t(y)
    assert np.all(np.abs(got - ref) <= 1e-13 * _spmv_abs_bound(A, x) + 1e-300)
    assert np.all(np.abs(got_t - ref_t) <= 1e-13 * _spmv_abs_bound(A.T, y) + 1e-300)
    # short rows stay bit-exact even when long rows exist
    short = np.diff(A.tocsr().indptr) <= 2048
    assert np.array_equal(got[short], ref[short])


def test_spmv_empty_matrix_and_empty_rows(gpu_required):
    import scipy.sparse as sp
    from firstorderlp_jl_amd import linear_programming_problem
    A = sp.csc_matrix((5, 4))
    p = linear_programming_problem(np.zeros(4), np.ones(4), np.ones(4), 0.0, A,
                                   np.ones(5), 2)
    eng = HipPdhgEngine.from_problem(p)
    assert np.array_equal(eng.spmv(np.ones(4)), np.zeros(5))
    assert np.array_equal(eng.spmv_t(np.ones(5)), np.zeros(4))
    raw = eng.trial_step(0.5, 1.0, 1.0)
    x1, y1, a1 = eng.get_trial()
    assert np.array_equal(x1, np.zeros(4))           # x - .5*c projected to [0,1]
    assert np.array_equal(y1[:2], 0.5 * np.ones(2))  # equality rows: y + .5*b
    assert raw[0] == 0.0


@pytest.mark.parametrize("maker,seed", [
    (lambda: H.example_lp(), 0), (lambda: H.example_cc_lp(), 0),
    (lambda: H.example_lp_without_bounds(), 0),
    (lambda: random_lp(2000, 1500, 8, 11), 11),
    (lambda: random_lp(30000, 40000, 10, 12), 12),
    (lambda: H.skewed_lp(2500, 7000, 13), 13)])
def test_trial_step_matches_oracle(gpu_required, maker, seed):
    p = maker()
    eng, st = _mk(p)
    m, n = p.constraint_matrix.shape
    rng = np.random.default_rng(seed)
    # start from a non-trivial state so every term is exercised
    x0 = np.clip(rng.standard_normal(n), np.maximum(p.variable_lower_bound, -5),
                 np.minimum(p.variable_upper_bound, 5))
    y0 = rng.standard_normal(m)
    y0[p.num_equalities:] = np.abs(y0[p.num_equalities:])
    eng.set_current(x0, y0)
    st.x, st.y = x0, y0
    st.recompute_dual_product()
    short_only = max(eng.layout_info()["A_max_row_nnz"],
                     eng.layout_info()["At_max_row_nnz"]) <= 2048
    if short_only:
        assert np.array_equal(eng.get_dual_product(), st.aty)
    step, pw = H.initial_step_and_weight(p)
    for theta in (1.0, 0.37):
        raw = eng.trial_step(step, pw, theta)
        raw_o, xn, yn, an = st.trial_step(step, pw, theta)
        gx, gy, ga = eng.get_trial()
        if short_only:
            assert np.array_equal(gx, xn)
            assert np.array_equal(gy, yn)
            assert np.array_equal(ga, an)
        else:
            np.testing.assert_allclose(gx, xn, rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(gy, yn, rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(ga, an, rtol=1e-11, atol=1e-11)
        scale = np.array([np.abs(raw_o[1] * raw_o[2]) ** 0.5 + abs(raw_o[0]),
                          raw_o[1], raw_o[2]])
        assert np.all(np.abs(raw[:3] - raw_o[:3]) <= 1e-12 * scale + 1e-300)
        assert abs(raw[3] - raw_o[3]) <= 1e-12 * raw_o[3] + 1e-300
        assert raw[4] == 0.0


@pytest.mark.parametrize("policy", ["adaptive", "constant", "malitsky-pock"])
def test_trajectory_matches_oracle(gpu_required, policy):
    p = random_lp(4000, 5000, 10, 21)
    eng, st = _mk(p)
    step, pw = H.initial_step_and_weight(p)
    if policy == "constant":
        step = 0.01
    state = PdhgSolverState(eng, step_size=step, primal_weight=pw,
                            ratio_step_sizes=1.0)
    st.step_size, st.primal_weight, st.ratio_step_sizes = step, pw, 1.0
    params = {"adaptive": AdaptiveStepsizeParams(0.3, 0.6),
              "constant": ConstantStepsizeParams(),
              "malitsky-pock": MalitskyPockStepsizeParameters(0.7, 0.99, 1.0)}[policy]
    for it in range(60):
        take_step(params, state)
        if policy == "adaptive":
            st.take_step_adaptive(0.3, 0.6)
        elif policy == "constant":
            st.take_step_constant()
        else:
            st.take_step_malitsky_pock(0.7, 0.99, 1.0)
    assert state.total_number_iterations == st.total_number_iterations
    assert state.step_size == pytest.approx(st.step_size, rel=1e-9)
    x, y = eng.get_current()
    np.testing.assert_allclose(x, st.x, rtol=1e-9, atol=1e-9 * np.abs(st.x).max())
    np.testing.assert_allclose(y, st.y, rtol=1e-9, atol=1e-9 * np.abs(st.y).max())
    xa, ya = eng.get_average()
    xo, yo = st.compute_average()
    np.testing.assert_allclose(xa, xo, rtol=1e-9, atol=1e-9 * np.abs(xo).max())
    np.testing.assert_allclose(ya, yo, rtol=1e-9, atol=1e-9 * np.abs(yo).max())
    cx, cy, wx, wy = eng.average_info()
    ocx, ocy, owx, owy = st.average_counts()
    assert (cx, cy) == (ocx, ocy)
    assert wx == pytest.approx(owx, rel=1e-9) and wy == pytest.approx(owy, rel=1e-9)


def test_restart_to_average_and_reset(gpu_required):
    p = random_lp(3000, 2000, 6, 31)
    eng, st = _mk(p)
    step, pw = H.initial_step_and_weight(p)
    state = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    st.step_size, st.primal_weight = step, pw
    for _ in range(10):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), state)
        st.take_step_adaptive(0.3, 0.6)
    xa, ya = eng.get_average()
    eng.restart_to_average()
    x, y = eng.get_current()
    assert np.array_equal(x, xa) and np.array_equal(y, ya)
    A = p.constraint_matrix
    assert np.array_equal(eng.get_dual_product(),
                          orc.spmv_t(A.shape[0], A.shape[1], A.indptr, A.indices, A.data, ya))
    eng.reset_average()
    assert eng.average_info() == (0, 0, 0.0, 0.0)


def test_adjoint_identity_large(gpu_required):
    """Size-independent property tying K3 and K5 together: <Ax, y> == <x, A'y>."""
    p = random_lp(1_000_000, 1_000_000, 10, 41)
    eng = HipPdhgEngine.from_problem(p)
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(eng.n), rng.standard_normal(eng.m)
    ax, aty = eng.spmv(x), eng.spmv_t(y)
    lhs, rhs = float(ax @ y), float(x @ aty)
    assert abs(lhs - rhs) <= 1e-10 * (np.linalg.norm(ax) * np.linalg.norm(y))
    A = p.constraint_matrix
    assert np.array_equal(ax, orc.spmv(eng.m, eng.n, A.indptr, A.indices, A.data, x))
    assert np.array_equal(aty, orc.spmv_t(eng.m, eng.n, A.indptr, A.indices, A.data, y))


def test_pagerank_lp_matches_oracle(gpu_required):
    """BASELINE configs[2] model at reduced size: one dense equality row (long
    row in A), hub columns (long rows in A'), objective 0."""
    from firstorderlp_jl_amd.generators import pagerank_lp
    p = pagerank_lp(60000, seed=1)
    eng, st = _mk(p)
    info = eng.layout_info()
    assert info["A_long_rows"] >= 1 and info["A_max_row_nnz"] == 60000
    step, pw = H.initial_step_and_weight(p)
    assert pw == 1.0                      # c == 0 -> primal_importance (saddle_point.jl:1058-1070)
    state = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    st.step_size, st.primal_weight = step, pw
    for _ in range(40):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), state)
        st.take_step_adaptive(0.3, 0.6)
    assert state.total_number_iterations == st.total_number_iterations
    x, y = eng.get_current()
    np.testing.assert_allclose(x, st.x, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(st.x).max()))
    np.testing.assert_allclose(y, st.y, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(st.y).max()))


def test_l1_svm_lp_matches_oracle(gpu_required):
    """BASELINE configs[3] model at reduced size: free variables, all-inequality
    rows, a dense intercept column (long row in A')."""
    from firstorderlp_jl_amd.generators import l1_svm_rcv1_like_lp
    p = l1_svm_rcv1_like_lp(num_samples=3000, num_features=4000, nnz_per_row=30, seed=2)
    eng, st = _mk(p)
    assert eng.layout_info()["At_long_rows"] >= 1
    step, pw = H.initial_step_and_weight(p)
    state = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    st.step_size, st.primal_weight = step, pw
    for _ in range(40):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), state)
        st.take_step_adaptive(0.3, 0.6)
    assert state.total_number_iterations == st.total_number_iterations
    x, y = eng.get_current()
    np.testing.assert_allclose(x, st.x, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(st.x).max()))
    np.testing.assert_allclose(y, st.y, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(st.y).max()))
