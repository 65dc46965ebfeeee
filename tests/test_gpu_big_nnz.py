"""nnz beyond the 32-bit layout limit (the reference's indices are Int64, quadratic_programming.jl:64).  Since round 4
pdhg_create holds such a matrix as SEGMENTS of whole rows inside one ordinary handle -- CSR(A) cut by rows, CSR(A') by
columns, 32-bit offsets local to a segment: no exchange, every row sum in its reference order, so both products stay
bit-exact with the oracle and whole trajectories bitwise those of the matrix in one piece.  PDHG_HUGE=shards keeps the
earlier form (row shards on the one device behind a group handle).  The limit is lowered through PDHG_MAX_SHARD_NNZ so
that both paths run on a small problem; RUN_HUGE_NNZ=1 additionally builds a real 2.2 G-nonzero matrix (tens of GB of
host memory, minutes)."""
import os

import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (AdaptiveStepsizeParams, PdhgSolverState,
                                                             take_step)
from oracle import oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("maker", [lambda: random_lp(60_000, 50_000, 10, seed=12),                      # stream layouts
                                   lambda: H.skewed_lp(30_000, 40_000, seed=3, dense_rows=2, dense_cols=2, base_nnz=8),   # + long rows
                                   lambda: random_lp(700_000, 650_000, 4, seed=101)],                # swept segments
                         ids=["stream", "long_rows", "tiled"])
def test_matrix_above_the_entry_limit_is_held_as_row_segments(gpu_required, monkeypatch, maker):
    p = maker()
    A = p.constraint_matrix
    m, n = A.shape
    monkeypatch.setenv("PDHG_MAX_SHARD_NNZ", str(A.nnz // 4))
    eng = HipPdhgEngine.from_problem(p)
    info = eng.layout_info()
    assert eng.dist_info()["world"] == 1 and info["A_segments"] >= 4 and info["At_segments"] >= 4, info
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    H.assert_products_match_oracle(eng, A, x, y, label="segmented")          # BOTH products: rows are whole in a segment
    monkeypatch.delenv("PDHG_MAX_SHARD_NNZ")
    one = HipPdhgEngine.from_problem(p)
    assert one.layout_info()["A_segments"] == 0
    assert np.array_equal(eng.spmv(x), one.spmv(x)) and np.array_equal(eng.spmv_t(y), one.spmv_t(y))
    outs = []
    for e in (eng, one):
        step, pw = H.initial_step_and_weight(p)
        st = PdhgSolverState(e, step_size=step, primal_weight=pw)
        for _ in range(50):
            take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
        outs.append((*e.get_current(), *e.get_average(), np.array([st.step_size]), st.total_number_iterations))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)               # exactly rounded sums: the block grouping of the segments does not show
    # device rescaling walks every segment (Ruiz: exact maxima, so the same factors and the same scaled entries)
    for e in (eng, one):
        e.rescale(10, False, 1.0)
    assert np.array_equal(eng.spmv(x), one.spmv(x)) and np.array_equal(eng.spmv_t(y), one.spmv_t(y))
    assert eng.matrix_max_abs() == one.matrix_max_abs()
    eng.close()
    one.close()


def test_matrix_above_the_shard_limit_is_cut_on_one_device(gpu_required, monkeypatch):
    p = random_lp(60_000, 50_000, 10, seed=12)           # 600k nonzeros
    A = p.constraint_matrix
    m, n = A.shape
    monkeypatch.setenv("PDHG_MAX_SHARD_NNZ", "150000")
    monkeypatch.setenv("PDHG_HUGE", "shards")            # rounds 2-3's form
    eng = HipPdhgEngine.from_problem(p)
    info = eng.dist_info()
    assert info["world"] == info["local_ranks"] == 5 and info["backend"] == 1     # 600k / (0.8 * 150k)
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    assert np.array_equal(eng.spmv(x), orc.spmv(m, n, A.indptr, A.indices, A.data, x))     # rows are whole in a shard
    np.testing.assert_allclose(eng.spmv_t(y), A.T @ y, rtol=1e-11, atol=1e-11)
    monkeypatch.delenv("PDHG_MAX_SHARD_NNZ")
    one = HipPdhgEngine.from_problem(p)
    assert one.dist_info()["world"] == 1
    outs = []
    for e in (eng, one):
        step, pw = H.initial_step_and_weight(p)
        st = PdhgSolverState(e, step_size=step, primal_weight=pw)
        for _ in range(50):
            take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
        outs.append((*e.get_current(), st.total_number_iterations))
    assert outs[0][2] == outs[1][2]
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-9, atol=1e-9)


@pytest.mark.timeout(3000)
@pytest.mark.skipif(os.environ.get("RUN_HUGE_NNZ", "0") != "1", reason="2.2 G nonzeros: set RUN_HUGE_NNZ=1")
def test_2p2_billion_nonzeros(gpu_required):
    """Layout construction and one product checksum at nnz > 2^31.  Structured matrix, built
    directly in CSC: row i has 22 entries, at columns (7 i + 1000003 k) mod n with value
    1/(k+1), so A*1 is the same known number in every row and A'*1 in every column."""
    import time
    n = m = 100_000_000
    K = 22
    inv7 = pow(7, -1, n)
    indptr = np.arange(0, K * n + 1, K, dtype=np.int64)
    indices = np.empty(K * n, dtype=np.int64)
    data = np.empty(K * n, dtype=np.float64)
    kk = np.arange(K, dtype=np.int64)
    t0 = time.time()
    for j0 in range(0, n, 5_000_000):
        j = np.arange(j0, min(n, j0 + 5_000_000), dtype=np.int64)
        rows = (((j[:, None] - 1000003 * kk[None, :]) % n) * inv7) % n      # 7 i = j - 1000003 k (mod n)
        order = np.argsort(rows, axis=1)
        indices[j0 * K:(j0 + len(j)) * K] = np.take_along_axis(rows, order, axis=1).reshape(-1)
        data[j0 * K:(j0 + len(j)) * K] = (1.0 / (order + 1.0)).reshape(-1)
    assert len(data) > 2 ** 31
    print(f"generated {len(data)} nonzeros in {time.time() - t0:.0f} s", flush=True)

    class Csc:            # what HipPdhgEngine reads of a scipy CSC matrix
        shape = (m, n)
        nnz = len(data)
    Csc.indptr, Csc.indices, Csc.data = indptr, indices, data
    t0 = time.time()
    eng = HipPdhgEngine(Csc, np.ones(n), np.ones(m), np.zeros(n), np.full(n, np.inf), 0)
    print(f"pdhg_create: {time.time() - t0:.0f} s, {eng.dist_info()}, {eng.layout_info()}", flush=True)
    assert eng.dist_info()["world"] == 1 and eng.layout_info()["A_segments"] >= 2 and eng.layout_info()["At_segments"] >= 2
    want = float(np.sum(1.0 / (kk + 1.0)))
    np.testing.assert_allclose(eng.spmv(np.ones(n)), np.full(m, want), rtol=1e-12)
    np.testing.assert_allclose(eng.spmv_t(np.ones(m)), np.full(n, want), rtol=1e-12)
    raw = eng.trial_step(0.1, 1.0, 1.0)
    assert np.all(np.isfinite(raw))
    eng.close()


def test_sharded_create_checks_every_shard_and_refuses_a_caller_stream(gpu_required, monkeypatch):
    """The partition works on whole rows: a heavy row block can exceed the limit although the
    average shard does not -- the shard count is raised until every shard fits; a single row
    beyond the limit is refused naming the row; a caller stream cannot be honoured."""
    import scipy.sparse as sp
    from firstorderlp_jl_amd.quadratic_programming import QuadraticProgrammingProblem
    rng = np.random.default_rng(5)
    m, n = 4000, 3000
    A = sp.random(m, n, density=0.004, format="lil", random_state=7)
    for r in range(40):                              # 40 rows of 1500 entries at the top: 60k of ~108k nonzeros
        cols = np.sort(rng.choice(n, 1500, replace=False))
        A.rows[r] = cols.tolist()
        A.data[r] = rng.standard_normal(1500).tolist()
    A = A.tocsc()
    p = QuadraticProgrammingProblem(
        variable_lower_bound=np.zeros(n), variable_upper_bound=np.full(n, np.inf),
        objective_matrix=sp.csc_matrix((n, n)), objective_vector=rng.standard_normal(n), objective_constant=0.0,
        constraint_matrix=A, right_hand_side=rng.standard_normal(m), num_equalities=100)
    monkeypatch.setenv("PDHG_MAX_SHARD_NNZ", "30000")
    monkeypatch.setenv("PDHG_HUGE", "shards")
    eng = HipPdhgEngine.from_problem(p)
    info = eng.dist_info()
    assert info["world"] >= 4 and info["backend"] == 1
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    np.testing.assert_allclose(eng.spmv(x), A @ x, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(eng.spmv_t(y), A.T @ y, rtol=1e-11, atol=1e-11)
    eng.close()
    monkeypatch.setenv("PDHG_MAX_SHARD_NNZ", "1000")            # a 1500-entry row cannot be indexed
    with pytest.raises(Exception, match="row 0 alone holds 1500"):
        HipPdhgEngine.from_problem(p)
    monkeypatch.delenv("PDHG_HUGE")                              # ... by a segment either
    with pytest.raises(Exception, match="row 0 alone holds 1500"):
        HipPdhgEngine.from_problem(p)
    monkeypatch.setenv("PDHG_HUGE", "shards")
    monkeypatch.setenv("PDHG_MAX_SHARD_NNZ", "30000")
    import torch
    with pytest.raises(Exception, match="caller-supplied stream"):
        HipPdhgEngine.from_problem(p, stream=torch.cuda.Stream().cuda_stream)


def test_segmented_matrix_through_a_whole_optimize(gpu_required, monkeypatch):
    """optimize() -- device rescaling, the evaluation branch (unscaled statistics, trust-region bounds), restarts to the
    average, termination -- on a matrix held as row segments: the same iteration count, termination reason and solution as
    the matrix in one piece (every kernel of the evaluation branch and of the rescaling walks the segments)."""
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import PdhgParameters, optimize
    from firstorderlp_jl_amd.saddle_point import RestartScheme, RestartToCurrentMetric, construct_restart_parameters
    from firstorderlp_jl_amd.termination import construct_termination_criteria
    q = random_lp(4000, 3000, 7, seed=23)
    tc = construct_termination_criteria(eps_optimal_absolute=1e-6, eps_optimal_relative=1e-6, iteration_limit=3000)
    rp = construct_restart_parameters(RestartScheme.ADAPTIVE_NORMALIZED, RestartToCurrentMetric.GAP_OVER_DISTANCE_SQUARED,
                                      1000, 0.5, 0.1, 0.9, 0.5, False)
    params = PdhgParameters(10, False, 1.0, 1.0, True, 0, True, 40, tc, rp, AdaptiveStepsizeParams(0.3, 0.6))
    outs = []
    for cap in (None, str(q.constraint_matrix.nnz // 5)):
        if cap is None:
            monkeypatch.delenv("PDHG_MAX_SHARD_NNZ", raising=False)
        else:
            monkeypatch.setenv("PDHG_MAX_SHARD_NNZ", cap)
        o = optimize(params, q)
        outs.append((o.iteration_count, o.termination_reason, o.primal_solution, o.dual_solution))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1], (outs[0][:2], outs[1][:2])
    assert np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][3], outs[1][3])
