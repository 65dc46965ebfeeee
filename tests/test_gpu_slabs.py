"""Column-slab passes of the stream layout (spmv_stream_kernel<MODE, INIT>): when the
gathered vector is a few times an XCD's L2 the product runs as one launch per column
slab, the row sums travelling through a partial buffer.  Rows receive their products in
the same left-to-right order as in a single pass, so everything must stay BIT-IDENTICAL
to the oracle -- including rows that are empty in a slab, long rows (kept whole on the
long-row path) and the fused epilogues -- and to the single-pass layout."""
import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import pagerank_lp, random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (AdaptiveStepsizeParams, PdhgSolverState,
                                                             take_step)
from oracle import oracle as orc
from tests import helpers as H

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def _with_dense_rows_and_columns(m, n, k, seed):
    """random_lp plus two dense rows and two dense columns (long rows in both layouts)."""
    import scipy.sparse as sp
    from firstorderlp_jl_amd import linear_programming_problem
    p = random_lp(m, n, k, seed)
    rng = np.random.default_rng(seed + 1)
    A = p.constraint_matrix.tocsr()
    A = sp.vstack([sp.csr_matrix(rng.standard_normal((2, n))), A[2:]]).tocsc()
    A = sp.hstack([sp.csc_matrix(rng.standard_normal((m, 2))), A[:, 2:]]).tocsc()
    A.sort_indices()
    return linear_programming_problem(p.variable_lower_bound, p.variable_upper_bound, p.objective_vector, 0.0,
                                      A, p.right_hand_side, p.num_equalities)


def _engine(p, monkeypatch, slab_mb, graph="1"):
    monkeypatch.setenv("PDHG_SPMV", "stream")
    monkeypatch.setenv("PDHG_GRAPH", graph)
    if slab_mb is None:
        monkeypatch.setenv("PDHG_SLABS", "0")
    else:
        monkeypatch.setenv("PDHG_SLABS", "1")
        monkeypatch.setenv("PDHG_SLAB_MB", str(slab_mb))
    return HipPdhgEngine.from_problem(p)


@pytest.mark.parametrize("maker,slab_mb", [(lambda: random_lp(200_000, 150_000, 8, seed=3), 0.5),
                                           (lambda: pagerank_lp(120_000, seed=4), 0.3),
                                           (lambda: _with_dense_rows_and_columns(150_000, 140_000, 6, seed=7), 0.4)],
                         ids=["random", "pagerank", "skewed_long_rows"])
def test_slab_passes_are_bit_identical(gpu_required, monkeypatch, maker, slab_mb):
    p = maker()
    A = p.constraint_matrix
    m, n = A.shape
    eng = _engine(p, monkeypatch, slab_mb)
    info = eng.layout_info()
    assert 2 <= info["A_slabs"] <= 4 and 2 <= info["At_slabs"] <= 4, info
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    ref, ref_t = orc.spmv(m, n, A.indptr, A.indices, A.data, x), orc.spmv_t(m, n, A.indptr, A.indices, A.data, y)
    got, got_t = eng.spmv(x), eng.spmv_t(y)
    short, short_t = np.diff(A.tocsr().indptr) <= H.bitexact_row_limit(), np.diff(A.indptr) <= H.bitexact_row_limit()
    assert np.array_equal(got[short], ref[short]) and np.array_equal(got_t[short_t], ref_t[short_t])
    assert np.all(np.abs(got - ref) <= 1e-13 * (abs(A) @ np.abs(x)) + 1e-300)
    # trajectories: slabs (graph and plain launches) and the single-pass layout
    def run(e):
        step, pw = H.initial_step_and_weight(p)
        st = PdhgSolverState(e, step_size=step, primal_weight=pw)
        for _ in range(40):
            take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
        return (*e.get_current(), *e.get_average(), st.step_size, st.total_number_iterations)
    r_slab = run(eng)
    r_slab_plain = run(_engine(p, monkeypatch, slab_mb, graph="0"))
    r_single = run(_engine(p, monkeypatch, None))
    for a, b, c in zip(r_slab, r_slab_plain, r_single):
        assert np.array_equal(a, b)                       # graph launch vs plain launches of the same layout
        # vs the single-pass layout the vectors are bit-identical per step, but the row blocks
        # (hence the grouping of the block partials of the three step scalars) differ
        np.testing.assert_allclose(a, c, rtol=1e-9, atol=1e-9)


def test_slab_layout_rescales_in_place(gpu_required, monkeypatch):
    p = random_lp(200_000, 150_000, 8, seed=5)
    slab, single = _engine(p, monkeypatch, 0.5), _engine(p, monkeypatch, None)
    assert slab.layout_info()["A_slabs"] >= 2 and single.layout_info()["A_slabs"] == 0
    for e in (slab, single):
        e.rescale(10, False, 1.0)
    rng = np.random.default_rng(2)
    x, y = rng.standard_normal(150_000), rng.standard_normal(200_000)
    assert np.array_equal(slab.spmv(x), single.spmv(x)) and np.array_equal(slab.spmv_t(y), single.spmv_t(y))


def test_default_thresholds(gpu_required):
    """Default: slabs of one XCD L2 (4 MiB) for vectors of 1.25 .. 4 slabs; PageRank-1M is the case in point."""
    small = HipPdhgEngine.from_problem(random_lp(5000, 4000, 8, seed=7)).layout_info()
    assert small["A_slabs"] == 0 and small["At_slabs"] == 0


def test_no_slab_passes_for_banded_rows(gpu_required, monkeypatch):
    """A banded matrix whose gathered vector (700 000 doubles: 5.3 MiB) is in the slab range: the sweep is declined because
    a few thousand consecutive rows touch a sliver of the columns -- and for the same reason the stream layout takes NO
    slab passes (layout.hpp: slab_count; 1M x 1M +-5 000 ran 2x slower than the vendor kernel with them).  PDHG_SLABS=2
    forces the passes: same bits either way, and the same as the oracle."""
    import scipy.sparse as sp
    from firstorderlp_jl_amd import linear_programming_problem
    monkeypatch.delenv("PDHG_SPMV", raising=False)
    monkeypatch.delenv("PDHG_SLABS", raising=False)
    m = n = 700_000
    rng = np.random.default_rng(12)
    rows = np.repeat(np.arange(m), 3)
    cols = np.clip(rows + rng.integers(-2000, 2001, rows.size), 0, n - 1)
    A = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(m, n))
    A.sum_duplicates()
    p = linear_programming_problem(np.zeros(n), np.full(n, 10.0), rng.standard_normal(n), 0.0, A.tocsc(),
                                   rng.standard_normal(m), m // 2)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    auto = HipPdhgEngine.from_problem(p)
    info = auto.layout_info()
    assert info["A_tiled_waves"] == 0 and info["At_tiled_waves"] == 0, info      # (banded: streamed)
    assert info["A_slabs"] == 0 and info["At_slabs"] == 0, info
    H.assert_products_match_oracle(auto, p.constraint_matrix, x, y)
    monkeypatch.setenv("PDHG_SLABS", "2")
    forced = HipPdhgEngine.from_problem(p)
    finfo = forced.layout_info()
    assert finfo["A_slabs"] == 2 and finfo["At_slabs"] == 2, finfo
    assert np.array_equal(forced.spmv(x), auto.spmv(x)) and np.array_equal(forced.spmv_t(y), auto.spmv_t(y))
    forced.close()
    auto.close()
    # rows that scatter over the same vector keep their slab passes (stream layout forced: the builder would sweep them)
    monkeypatch.setenv("PDHG_SLABS", "1")
    monkeypatch.setenv("PDHG_SPMV", "stream")
    scattered = HipPdhgEngine.from_problem(random_lp(300_000, 700_000, 5, seed=3)).layout_info()
    assert scattered["A_slabs"] == 2, scattered
