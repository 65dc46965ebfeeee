"""PDHG_ROW_ORDER: "strict" adds every row's products left to right on one lane (bit-exact with
the CPU oracle's sequential loops for rows of <= 2048 entries); "relaxed" -- the library's
default -- sums rows of more than 256 entries (stream layout) / same-row runs of more than 8
entries inside a tile (sweep, only when forced onto a matrix with runs beyond 32) wave-parallel in a FIXED order: reproducible, rows of <= 256
entries still bit-exact in the stream layout, everything within 1e-13 * sum |a_ij x_j| of the
sequential sum -- the bar rows beyond 2048 entries have always had (saddle_point.jl:1102-1107,
pdhg.jl:472-494 are the products)."""
import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import l1_svm_rcv1_like_lp, pagerank_lp, random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (AdaptiveStepsizeParams, PdhgSolverState,
                                                             take_step)
from oracle import oracle as orc
from tests import helpers as H
from tests import kat_common

pytestmark = pytest.mark.gpu

MAKERS = {
    "pagerank": lambda: pagerank_lp(60000, seed=2),
    "skewed": lambda: H.skewed_lp(6000, 9000, seed=7, dense_rows=3, dense_cols=3),
    "l1svm": lambda: l1_svm_rcv1_like_lp(num_samples=3000, num_features=6000, nnz_per_row=40, seed=0),
    "random": lambda: random_lp(20000, 15000, 9, seed=5),
}


def _products(p, monkeypatch, order, layout):
    monkeypatch.setenv("PDHG_ROW_ORDER", order)
    monkeypatch.setenv("PDHG_SPMV", layout)
    if layout == "tiled":
        monkeypatch.setenv("PDHG_TILE_COLS", "1024")
    eng = HipPdhgEngine.from_problem(p)
    m, n = p.constraint_matrix.shape
    rng = np.random.default_rng(11)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    out = (eng.spmv(x), eng.spmv_t(y), eng.layout_info())
    eng.close()
    return out, x, y


@pytest.mark.parametrize("layout", ["stream", "tiled"])
@pytest.mark.parametrize("name", sorted(MAKERS))
def test_relaxed_products_are_within_the_stated_bound_and_short_rows_bit_exact(gpu_required, monkeypatch, name, layout):
    p = MAKERS[name]()
    A = p.constraint_matrix
    m, n = A.shape
    (ax_r, aty_r, info_r), x, y = _products(p, monkeypatch, "relaxed", layout)
    (ax_s, aty_s, _), _, _ = _products(p, monkeypatch, "strict", layout)
    ref, ref_t = orc.spmv(m, n, A.indptr, A.indices, A.data, x), orc.spmv_t(m, n, A.indptr, A.indices, A.data, y)
    absA = abs(A)
    bound, bound_t = 1e-13 * (absA @ np.abs(x)), 1e-13 * (absA.T @ np.abs(y))
    assert np.all(np.abs(ax_r - ref) <= bound + 1e-300) and np.all(np.abs(aty_r - ref_t) <= bound_t + 1e-300)
    len_rows, len_cols = np.diff(A.tocsr().indptr), np.diff(A.indptr)
    # strict: bit-exact for every row the long-row path does not take
    assert np.array_equal(ax_s[len_rows <= 2048], ref[len_rows <= 2048])
    assert np.array_equal(aty_s[len_cols <= 2048], ref_t[len_cols <= 2048])
    if layout == "stream":
        # relaxed, stream layout: rows of <= 256 entries are still added left to right by one lane
        assert np.array_equal(ax_r[len_rows <= 256], ref[len_rows <= 256])
        assert np.array_equal(aty_r[len_cols <= 256], ref_t[len_cols <= 256])
    else:
        # relaxed sweep: runs of <= 8 entries inside a tile are strict; rows of <= 8 entries always are
        assert np.array_equal(ax_r[len_rows <= 8], ref[len_rows <= 8])
        assert np.array_equal(aty_r[len_cols <= 8], ref_t[len_cols <= 8])
    if name != "random":
        assert max(len_rows.max(), len_cols.max()) > 256         # the test matrices do have wide rows


def test_hub_matrices_keep_the_stream_layout_in_both_orders(gpu_required, monkeypatch):
    """A PageRank graph large enough for the automatic layout choice to try the sweep (gathered
    vector beyond 3 MiB): its hub rows' long same-row runs make build_tiled decline in strict
    order (one lane would add hundreds of products per tile) and -- measured, 0.168 ms swept
    against 0.104 ms streamed on PageRank-1M -- in relaxed order too; forcing the sweep in relaxed
    order selects the shuffle-tree chunk variant and stays within the stated bound (test above)."""
    p = pagerank_lp(600_000, seed=3)
    monkeypatch.delenv("PDHG_SPMV", raising=False)
    for order in ("strict", "relaxed"):
        monkeypatch.setenv("PDHG_ROW_ORDER", order)
        eng = HipPdhgEngine.from_problem(p)
        info = eng.layout_info()
        eng.close()
        assert info["A_tiled_waves"] == 0 and info["At_tiled_waves"] == 0, (order, info)


@pytest.mark.parametrize("path", ["plain", "graph", "one_kernel"])
@pytest.mark.parametrize("name", ["pagerank", "skewed"])
def test_relaxed_trajectories_follow_the_oracle_and_are_reproducible(gpu_required, monkeypatch, name, path):
    monkeypatch.setenv("PDHG_ROW_ORDER", "relaxed")
    monkeypatch.setenv("PDHG_GRAPH", "0" if path == "plain" else "1")
    monkeypatch.setenv("PDHG_COOP", "1" if path == "one_kernel" else "0")
    p = MAKERS[name]()
    runs = []
    for _ in range(2):
        eng = HipPdhgEngine.from_problem(p)
        step, pw = H.initial_step_and_weight(p)
        st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
        for _ in range(30):
            take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
        runs.append((np.concatenate(eng.get_current()), st.total_number_iterations, st.step_size))
        eng.close()
    assert np.array_equal(runs[0][0], runs[1][0]) and runs[0][1:] == runs[1][1:]      # fixed order: bitwise reproducible
    o = H.oracle_from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    o.step_size, o.primal_weight = step, pw
    for _ in range(30):
        o.take_step_adaptive(0.3, 0.6)
    assert o.total_number_iterations == runs[0][1]
    np.testing.assert_allclose(runs[0][0], np.concatenate([o.x, o.y]), rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("case", kat_common.CASES, ids=lambda c: c.__name__)
def test_reference_kats_in_relaxed_order_through_the_one_kernel_trial(gpu_required, monkeypatch, case):
    """The reference's 22 known-answer tests (test/test_primal_dual_hybrid_gradient.jl:77-423) on the
    library's DEFAULT configuration: relaxed row order, one persistent kernel per trial."""
    monkeypatch.setenv("PDHG_ROW_ORDER", "relaxed")
    monkeypatch.setenv("PDHG_GRAPH", "1")
    monkeypatch.setenv("PDHG_COOP", "1")
    case(lambda problem: HipPdhgEngine.from_problem(problem))
