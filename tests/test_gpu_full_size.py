"""BASELINE configs[4] at FULL size (m = n = 10M, nnz = 100M) on the GPU:
 * bit-exact A*x and A'*y against the oracle's sequential loops (tiled layout);
 * the adjoint identity <A x, y> = <x, A'y> tying K3 and K5 together;
 * linearity A(ax + bz) = a Ax + b Az to 1e-13 * |A||.|;
 * three adaptive PDHG steps: trial vectors bit-exact, scalars to 1e-12."""
import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import random_lp
from oracle import oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_config_s_full_size(gpu_required):
    m = n = 10_000_000
    p = random_lp(m, n, 10, 12345)
    A = p.constraint_matrix
    assert A.nnz == 100_000_000
    eng = HipPdhgEngine.from_problem(p)
    info = eng.layout_info()
    assert info["A_tiled_waves"] > 0 and info["At_tiled_waves"] > 0   # the v2 layout is the one under test
    rng = np.random.default_rng(0)
    x, z, y = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(m)
    ax, aty = eng.spmv(x), eng.spmv_t(y)
    assert np.array_equal(ax, orc.spmv(m, n, A.indptr, A.indices, A.data, x))
    assert np.array_equal(aty, orc.spmv_t(m, n, A.indptr, A.indices, A.data, y))
    lhs, rhs = float(ax @ y), float(x @ aty)
    assert abs(lhs - rhs) <= 1e-10 * np.linalg.norm(ax) * np.linalg.norm(y)
    az = eng.spmv(z)
    comb = eng.spmv(0.7 * x - 1.3 * z)
    bound = 0.7 * np.abs(ax) + 1.3 * np.abs(az) + 1.0
    assert np.all(np.abs(comb - (0.7 * ax - 1.3 * az)) <= 1e-12 * bound)

    st = H.oracle_from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    for it in range(3):
        raw = eng.trial_step(step, pw, 1.0)
        raw_o, xn, yn, an = st.trial_step(step, pw, 1.0)
        gx, gy, ga = eng.get_trial()
        assert np.array_equal(gx, xn) and np.array_equal(gy, yn) and np.array_equal(ga, an), it
        assert np.all(np.abs(raw[:4] - raw_o[:4]) <= 1e-12 * np.abs(raw_o[:4]) + 1e-300), (raw, raw_o)
        st.step_size = step
        st.accept(xn, yn, an)
        eng.accept(step)
        step *= 1.1
    xs, ys = eng.get_current()
    assert np.array_equal(xs, st.x) and np.array_equal(ys, st.y)


def _full_size_trial_check(p, label, steps=3):
    """Products and three adaptive trial steps against the oracle on a whole benchmark LP: rows up to the row order's
    bit-exact limit bitwise, longer rows within 1e-13 * sum |a x|; after the first accepted step the long rows' last-bit
    differences are in every vector, so the later steps are held to 1e-9."""
    A = p.constraint_matrix
    m, n = A.shape
    eng = HipPdhgEngine.from_problem(p)
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    absA = abs(A).tocsr()
    row_nnz, col_nnz = np.diff(A.tocsr().indptr), np.diff(A.indptr)
    H.assert_rows_match_oracle(eng.spmv(x), orc.spmv(m, n, A.indptr, A.indices, A.data, x), row_nnz, absA @ np.abs(x), label + " A x")
    H.assert_rows_match_oracle(eng.spmv_t(y), orc.spmv_t(m, n, A.indptr, A.indices, A.data, y), col_nnz, absA.T @ np.abs(y),
                               label + " A'y")
    st = H.oracle_from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    for it in range(steps):
        raw = eng.trial_step(step, pw, 1.0)
        raw_o, xn, yn, an = st.trial_step(step, pw, 1.0)
        gx, gy, ga = eng.get_trial()
        if it == 0:
            assert np.array_equal(gx, xn), label                      # elementwise from exact inputs
            sig = pw * step
            xbar = 2.0 * xn - st.x
            H.assert_rows_match_oracle(gy, yn, row_nnz, sig * (absA @ np.abs(xbar)) + np.abs(yn) + 1.0, label + " y'")
            H.assert_rows_match_oracle(ga, an, col_nnz, absA.T @ np.abs(yn) + 1.0, label + " A'y'")
        else:
            for g, o in ((gx, xn), (gy, yn), (ga, an)):
                assert np.allclose(g, o, rtol=1e-9, atol=1e-9 * (1.0 + np.abs(o).max())), (label, it)
        scale = np.abs(raw_o[:4]) + 1e-9 * np.abs(raw_o[:4]).max()
        assert np.all(np.abs(raw[:4] - raw_o[:4]) <= 1e-8 * scale + 1e-300), (label, it, raw, raw_o)
        st.step_size = step
        st.accept(xn, yn, an)
        eng.accept(step)
        step *= 1.1
    xs, ys = eng.get_current()
    assert np.allclose(xs, st.x, rtol=1e-9, atol=1e-9 * (1.0 + np.abs(st.x).max()))
    assert np.allclose(ys, st.y, rtol=1e-9, atol=1e-9 * (1.0 + np.abs(st.y).max()))
    info = eng.layout_info()
    eng.close()
    return info


@pytest.mark.timeout(900)
def test_pagerank_1m_full_size(gpu_required):
    """BASELINE configs[2] at the size bench.py runs it (1M nodes, ~10M nonzeros; column slabs + a 1M-entry long row)."""
    from firstorderlp_jl_amd.generators import pagerank_lp
    n = 1_000_000
    p = pagerank_lp(n, 4 * n, 0.99, seed=0)
    info = _full_size_trial_check(p, "pagerank-1M")
    assert info["A_long_rows"] >= 1 and info["A_max_row_nnz"] >= n      # the dense equality row takes the long-row path


@pytest.mark.timeout(900)
def test_l1svm_full_size(gpu_required):
    """BASELINE configs[3] SUBSTITUTE at the size bench.py runs it (20 242 x 47 236 rcv1-shaped data, 1.7M nonzeros)."""
    from firstorderlp_jl_amd.generators import l1_svm_rcv1_like_lp
    p = l1_svm_rcv1_like_lp(seed=0)
    info = _full_size_trial_check(p, "l1svm")
    assert info["At_long_chunks"] > 0                                    # dense feature columns: long rows of A'


@pytest.mark.own_row_order          # rows of config S have ~10 entries: both row orders take the same code path
@pytest.mark.timeout(1200)
def test_config_s_free_running_trajectory_is_bitwise_the_exact_sums_oracle(gpu_required):
    """BASELINE configs[4] at FULL size, free-running: 12 adaptive take_steps from the zero start (step-size rule, accepts,
    rejects and the weighted average included), the library through its batched call, the CPU oracle in exact-sums mode.
    Step sizes, iteration counts, iterates and averages must be the same BITS -- the north star's "same iterates within a
    stated tolerance" at tolerance zero on the headline LP (the oracle's default sequential sums differ from BLAS-order
    sums by construction; see DESIGN.md section 2 for what that comparison can and cannot show)."""
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_steps
    m = n = 10_000_000
    p = random_lp(m, n, 10, 12345)
    eng = HipPdhgEngine.from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    o = H.oracle_from_problem(p)
    o.exact_sums = True
    o.step_size, o.primal_weight = step, pw
    K = 12
    done = 0
    while done < K:
        done += take_steps(AdaptiveStepsizeParams(0.3, 0.6), st, K - done)
    for _ in range(K):
        o.take_step_adaptive(0.3, 0.6)
    assert st.total_number_iterations == o.total_number_iterations
    assert st.step_size == o.step_size
    x, y = eng.get_current()
    assert np.array_equal(x, o.x) and np.array_equal(y, o.y)
    xa, ya = eng.get_average()
    xo, yo = o.compute_average()
    assert np.array_equal(xa, xo) and np.array_equal(ya, yo)
    eng.close()
    o.close()
