"""The sliced jagged layout of the stream class (csrc/sj_kernels.hpp): rows sorted by length inside 2 048-row windows,
64 rows per wave stored level-major, every lane adding ITS row's products strictly left to right in a register.

* every row of at most 2 048 entries is BIT-IDENTICAL to the oracle's sequential loops (saddle_point.jl:1102-1107,
  pdhg.jl:492) -- in this layout also in the shipped relaxed row order, which only concerns the CSR row blocks;
* trajectories (accept / reject decisions, iterates, averages) equal the CSR row-block layout's: bitwise in strict order
  (both add every row left to right; the step sums are exactly rounded double-double sums), with plain launches and as
  a HIP graph;
* long rows (> 2 048 entries) stay with the long-row kernels, empty rows and a ragged last window are covered;
* column-slab passes (INIT carry) on the sliced jagged copies of the slabs;
* the builder picks the layout by itself for stream-class matrices with more than 1 024 row blocks."""
import numpy as np
import pytest
import scipy.sparse as sp

from firstorderlp_jl_amd import HipPdhgEngine, linear_programming_problem
from firstorderlp_jl_amd.generators import pagerank_lp, random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_step
from oracle import oracle as orc
from tests import helpers as H

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _ragged_lp(m, n, seed):
    """Row lengths from 0 to ~1 500 in no order, two rows and two columns beyond 2 048 entries, empty rows and columns."""
    rng = np.random.default_rng(seed)
    lens = np.minimum(rng.geometric(0.08, m), 1500)
    lens[rng.random(m) < 0.05] = 0
    lens[rng.integers(0, m, 40)] = rng.integers(300, 1500, 40)
    rows = np.repeat(np.arange(m), lens)
    cols = rng.integers(0, n - 50, rows.size)            # the last 50 columns stay empty
    A = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(m, n)).tolil()
    for r in (3, m - 2):
        A[r, :n - 50] = rng.standard_normal(n - 50)
    for c in (1, n - 60):
        A[:, c] = rng.standard_normal((m, 1))
    A = A.tocsc()
    A.sum_duplicates()
    A.sort_indices()
    return linear_programming_problem(np.zeros(n), np.full(n, 5.0), rng.standard_normal(n), 0.0, A, rng.standard_normal(m), m // 3)


def _engine(p, monkeypatch, sj, graph="1", slab_mb=None, wide=None):
    monkeypatch.setenv("PDHG_SPMV", "stream")
    if wide is None:
        monkeypatch.delenv("PDHG_SJ_WIDE", raising=False)
    else:
        monkeypatch.setenv("PDHG_SJ_WIDE", wide)         # dev: 0 = 256-row windows, 1 = 2 048-row windows (round 6)
    monkeypatch.setenv("PDHG_COOP", "0")                 # the products as their own kernels (graph nodes or plain launches)
    monkeypatch.setenv("PDHG_GRAPH", graph)
    monkeypatch.setenv("PDHG_SJ", sj)
    if slab_mb is None:
        monkeypatch.setenv("PDHG_SLABS", "0")
    else:
        monkeypatch.setenv("PDHG_SLABS", "1")
        monkeypatch.setenv("PDHG_SLAB_MB", str(slab_mb))
    return HipPdhgEngine.from_problem(p)


def _run(e, p, steps=40):
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(e, step_size=step, primal_weight=pw)
    for _ in range(steps):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
    return (*e.get_current(), *e.get_average(), e.get_dual_product(), st.step_size, st.total_number_iterations)


@pytest.mark.parametrize("maker", [lambda: random_lp(30_000, 20_000, 6, seed=21), lambda: _ragged_lp(5_000, 7_001, seed=2),
                                   lambda: H.skewed_lp(3_000, 9_000, seed=7, dense_rows=2, dense_cols=2),
                                   lambda: random_lp(70, 50, 3, seed=1)],
                         ids=["random", "ragged", "skewed_long_rows", "tiny"])
@pytest.mark.parametrize("wide", ["0", "1"], ids=["narrow", "wide"])
def test_products_are_bit_identical_to_the_oracle_in_every_row_order(gpu_required, monkeypatch, row_order_mode, maker, wide):
    """Rows a lane walks (<= 128 entries) are the oracle's bits in BOTH row orders; hub rows (129 ... 2 048 entries: whole-workgroup
    row blocks of the CSR arrays, round 6) follow the CSR kernel's rule -- bitwise up to 256 entries, and up to 2 048 in strict
    order; relaxed order sums rows beyond 256 entries by their wave (1e-13 * sum |a x|)."""
    p = maker()
    A = p.constraint_matrix
    m, n = A.shape
    eng = _engine(p, monkeypatch, "1", wide=wide)
    info = eng.layout_info()
    assert info["A_sj"] == 1 and info["At_sj"] == 1 and info["A_sj_wide"] == int(wide) and info["At_sj_wide"] == int(wide), info
    hubs = int(((np.diff(A.tocsr().indptr) > 128) & (np.diff(A.tocsr().indptr) <= 2048)).sum())
    assert info["A_sj_hub_rows"] == hubs, (info, hubs)
    exact_to = 2048 if row_order_mode == "strict" else 256
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    absA = abs(A).tocsr()
    for got, want, nnz_per, scale in ((eng.spmv(x), orc.spmv(m, n, A.indptr, A.indices, A.data, x), np.diff(A.tocsr().indptr), absA @ np.abs(x)),
                                      (eng.spmv_t(y), orc.spmv_t(m, n, A.indptr, A.indices, A.data, y), np.diff(A.indptr), absA.T @ np.abs(y))):
        short = nnz_per <= exact_to
        assert np.array_equal(got[short], want[short])
        assert np.all(np.abs(got - want) <= 1e-13 * scale + 1e-300)      # the long-row kernels
    # the fused products: one trial against the oracle's trial
    st = H.oracle_from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    raw = eng.trial_step(step, pw, 1.0)
    raw_o, xn, yn, an = st.trial_step(step, pw, 1.0)
    gx, gy, ga = eng.get_trial()
    assert np.array_equal(gx, xn)
    short_r, short_c = np.diff(A.tocsr().indptr) <= exact_to, np.diff(A.indptr) <= exact_to
    assert np.array_equal(gy[short_r], yn[short_r]) and np.allclose(gy, yn, rtol=1e-12, atol=1e-12)
    if short_r.all():
        assert np.array_equal(ga[short_c], an[short_c])
    assert np.allclose(raw[:4], raw_o[:4], rtol=1e-11, atol=1e-300)
    st.close()


@pytest.mark.parametrize("maker,wide", [(lambda: random_lp(30_000, 20_000, 6, seed=21), "1"), (lambda: _ragged_lp(5_000, 7_001, seed=2), "0"),
                                        (lambda: _ragged_lp(5_000, 7_001, seed=2), "1")], ids=["random-wide", "ragged-narrow", "ragged-wide"])
def test_trajectories_equal_the_csr_row_block_layout(gpu_required, monkeypatch, row_order_mode, maker, wide):
    p = maker()
    r_sj = _run(_engine(p, monkeypatch, "1", wide=wide), p)
    r_sj_plain = _run(_engine(p, monkeypatch, "1", graph="0", wide=wide), p)
    r_csr = _run(_engine(p, monkeypatch, "0"), p)
    for a, b, c in zip(r_sj, r_sj_plain, r_csr):
        assert np.array_equal(a, b)                       # graph nodes vs plain launches
        if row_order_mode == "strict" or np.diff(p.constraint_matrix.indptr).max() <= 256:
            assert np.array_equal(a, c)                   # both layouts add every row left to right
        else:
            np.testing.assert_allclose(a, c, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("maker,slab_mb", [(lambda: random_lp(200_000, 150_000, 8, seed=3), 0.5),
                                           (lambda: pagerank_lp(120_000, seed=4), 0.3)], ids=["random", "pagerank"])
@pytest.mark.parametrize("wide", ["0", "1"], ids=["narrow", "wide"])
@pytest.mark.strict_rows          # (the relaxed order only changes how hub rows beyond 256 entries are summed: covered by the products test)
def test_slab_passes_on_the_sliced_jagged_copies(gpu_required, monkeypatch, row_order_mode, maker, slab_mb, wide):
    p = maker()
    A = p.constraint_matrix
    m, n = A.shape
    eng = _engine(p, monkeypatch, "1", slab_mb=slab_mb, wide=wide)
    info = eng.layout_info()
    assert 2 <= info["A_slabs"] <= 4 and 2 <= info["At_slabs"] <= 4 and info["A_sj"] == 1 and info["At_sj"] == 1, info
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    ref, ref_t = orc.spmv(m, n, A.indptr, A.indices, A.data, x), orc.spmv_t(m, n, A.indptr, A.indices, A.data, y)
    got, got_t = eng.spmv(x), eng.spmv_t(y)
    exact_to = 2048 if row_order_mode == "strict" else 256       # hub rows (> 128 entries inside a slab) follow the CSR kernel's rule
    short, short_t = np.diff(A.tocsr().indptr) <= exact_to, np.diff(A.indptr) <= exact_to
    assert np.array_equal(got[short], ref[short]) and np.array_equal(got_t[short_t], ref_t[short_t])
    r_sj = _run(eng, p)
    r_csr = _run(_engine(p, monkeypatch, "0", slab_mb=slab_mb), p)
    for a, c in zip(r_sj, r_csr):
        if row_order_mode == "strict":
            assert np.array_equal(a, c)
        else:
            np.testing.assert_allclose(a, c, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("slab_mb", [None, 0.5], ids=["one_pass", "slabs"])
def test_device_rescaling_reaches_the_sliced_jagged_copies(gpu_required, monkeypatch, slab_mb):
    """pdhg_rescale scales every resident copy of the matrix: the sliced jagged copy must come out with the CSR copy's bits
    (Ruiz 10 + Pock-Chambolle, the reference's default preprocessing, preprocess.jl:412-573)."""
    p = random_lp(200_000, 150_000, 8, seed=5)
    sj, csr = _engine(p, monkeypatch, "1", slab_mb=slab_mb), _engine(p, monkeypatch, "0", slab_mb=slab_mb)
    assert sj.layout_info()["A_sj"] == 1 and csr.layout_info()["A_sj"] == 0
    for e in (sj, csr):
        e.rescale(10, False, 1.0)
    rng = np.random.default_rng(2)
    x, y = rng.standard_normal(150_000), rng.standard_normal(200_000)
    assert np.array_equal(sj.spmv(x), csr.spmv(x)) and np.array_equal(sj.spmv_t(y), csr.spmv_t(y))
    assert sj.matrix_max_abs() == csr.matrix_max_abs()


def test_the_builder_picks_the_layout_for_bandwidth_bound_stream_matrices(gpu_required, monkeypatch):
    """banded 1.5M x 1.5M, 8 per row: > 1 024 row blocks, rows that do not scatter -> stream class (three column slabs).  Both
    A (every row 8 entries, 7-8 after duplicate columns merge) and A' (column counts Poisson(8): a 256-row group is ~44 %
    full against its longest row) take the sliced jagged layout; a matrix with ragged groups does not."""
    monkeypatch.delenv("PDHG_SJ", raising=False)
    m = n = 1_500_000
    rng = np.random.default_rng(6)
    cols = np.clip(np.repeat(np.arange(m), 8) + rng.integers(-20_000, 20_001, m * 8), 0, n - 1)
    M = sp.csr_matrix((rng.standard_normal(m * 8), (np.repeat(np.arange(m), 8), cols)), shape=(m, n))
    M.sum_duplicates()
    p = linear_programming_problem(np.zeros(n), np.full(n, 10.0), rng.standard_normal(n), 0.0, M.tocsc(), rng.standard_normal(m), m // 2)
    eng = HipPdhgEngine.from_problem(p)
    info = eng.layout_info()
    assert info["A_tiled_waves"] == 0 and info["A_blocks"] > 1024 and info["A_sj"] == 1 and info["At_sj"] == 1, info
    # rows of one length keep the 256-row windows; Poisson column counts fill 2 048-row windows better (round 6)
    assert info["A_sj_wide"] == 0 and info["At_sj_wide"] == 1, (info, eng.layout_describe())
    assert "spmv_sj_kernel" in eng.kernel_name(1) and "spmv_sj_kernel" in eng.kernel_name(2)
    H.assert_products_match_oracle(eng, p.constraint_matrix, rng.standard_normal(n), rng.standard_normal(m), label="banded")
    r_auto = _run(eng, p, 20)
    monkeypatch.setenv("PDHG_SJ", "0")
    monkeypatch.setenv("PDHG_STREAM_PIPE", "0")
    r_csr = _run(HipPdhgEngine.from_problem(p), p, 20)
    for a, c in zip(r_auto, r_csr):
        assert np.array_equal(a, c)                       # rows of <= 30 entries: left to right in every layout and row order
    monkeypatch.delenv("PDHG_SJ", raising=False)
    ragged = HipPdhgEngine.from_problem(_ragged_lp(400_000, 300_000, seed=3)).layout_info()
    assert ragged["A_sj"] == 0, ragged                    # rows of up to 1 500 entries: not this layout
    small = HipPdhgEngine.from_problem(random_lp(5000, 4000, 8, seed=7)).layout_info()
    assert small["A_sj"] == 0 and small["At_sj"] == 0


@pytest.mark.short_rows
def test_device_rescaling_reaches_the_hessians_sliced_jagged_copies(gpu_required, monkeypatch):
    """A QP whose objective matrix runs `spmv_sj_kernel` (forced here; by itself for a sparse Hessian of n >~ 0.4M): after
    pdhg_rescale the sliced jagged copies of Q and Q' must hold (D^-1 Q) D^-1 (preprocess.jl:562-564) like the CSR
    arrays -- round 5 scaled only the CSR and slab arrays, so the products by Q used the UNSCALED Hessian (advisor r5,
    high).  Trial steps (x' = x - tau (c + Q x - A'y), pdhg.jl:462-477, and dx'Q dx) and 30 free-running steps bitwise the
    CSR layout's."""
    n, m = 12_000, 8_000
    p = random_lp(m, n, 6, seed=31)
    rng = np.random.default_rng(4)
    B = H.sparse_uniform(n, n, 3.0 / n, 9)
    p.objective_matrix = sp.csc_matrix(B.T @ B + sp.diags(rng.uniform(0.5, 3.0, n)))
    p.objective_matrix.sum_duplicates()
    p.objective_matrix.sort_indices()
    sj, csr = _engine(p, monkeypatch, "1"), _engine(p, monkeypatch, "0")
    assert sj.layout_info()["A_sj"] == 1 and csr.layout_info()["A_sj"] == 0
    assert sj.layout_info().get("Q_sj") == 1 and csr.layout_info().get("Q_sj") == 0
    x0, y0 = np.abs(rng.standard_normal(n)), rng.standard_normal(m)
    for e in (sj, csr):
        e.rescale(10, False, 1.0)
        e.set_current(x0, y0)
    ra, rb = sj.trial_step(0.05, 1.3), csr.trial_step(0.05, 1.3)
    assert np.array_equal(ra, rb)
    assert all(np.array_equal(u, v) for u, v in zip(sj.get_trial(), csr.get_trial()))
    for a, c in zip(_run(sj, p, 30), _run(csr, p, 30)):
        assert np.array_equal(a, c)


def test_power_law_rows_with_hub_rows_split_off(gpu_required, monkeypatch, row_order_mode):
    """PageRank LP (generate_pagerank_lp.jl:48-73), 300 000 nodes: rows of 5 ... ~2 000 entries by a power law and one
    dense row.  Round 6 can RUN it on the layout -- rows beyond 128 entries are hub rows (whole-workgroup row blocks inside
    the same kernel), the rest sorted inside 2 048-row windows -- and measured that it should not (0.22-0.37 ms per product
    on PageRank-1M against 0.10 on the CSR row blocks: a third of the lane-levels sit in slices of 30 ... 128 levels, walked at
    the latency of their batches), so the builder's rule declines it (`ragged_share`); forced, the products are the oracle's
    and 25 free-running steps the CSR row blocks'."""
    monkeypatch.delenv("PDHG_SJ", raising=False)
    monkeypatch.delenv("PDHG_SJ_WIDE", raising=False)
    p = pagerank_lp(300_000, seed=5)
    auto = HipPdhgEngine.from_problem(p)
    assert auto.layout_info()["A_sj"] == 0 and auto.layout_info()["At_sj"] == 0, auto.layout_describe()
    monkeypatch.setenv("PDHG_SJ", "1")
    eng = HipPdhgEngine.from_problem(p)
    info, desc = eng.layout_info(), eng.layout_describe()
    assert info["A_tiled_waves"] == 0 and info["A_sj"] == 1 and info["At_sj"] == 1, desc
    assert info["A_sj_wide"] == 1 and info["At_sj_wide"] == 1 and info["A_sj_hub_rows"] > 0 and info["A_long_rows"] == 1, desc
    sj = desc["A"]["sliced_jagged"]
    assert sj["fill_wide"] > sj["fill_narrow"] + 0.08 and sj["hub_threshold"] == 128 and sj["ragged_share"] > 0.10, sj
    assert "spmv_sj_kernel" in eng.kernel_name(1) and ", 8>" in eng.kernel_name(1)
    rng = np.random.default_rng(3)
    H.assert_products_match_oracle(eng, p.constraint_matrix, rng.standard_normal(eng.n), rng.standard_normal(eng.m), label="pagerank-300k")
    r_sj = _run(eng, p, 25)
    r_csr = _run(auto, p, 25)
    for a, c in zip(r_sj, r_csr):
        if row_order_mode == "strict":
            assert np.array_equal(a, c)
        else:
            np.testing.assert_allclose(a, c, rtol=1e-9, atol=1e-9)


def test_layout_describe_reports_the_timed_choices(gpu_required, monkeypatch):
    """pdhg_layout_describe (abi 11): the sweep's chunk variant / XCD dealing with the candidates pdhg_create timed on the
    matrix, and PDHG_TUNE=0 pinning the static rule (advisor r5: the dispatch of a handle must be reportable and pinnable)."""
    monkeypatch.setenv("PDHG_SPMV", "tiled")
    p = random_lp(400_000, 300_000, 10, seed=8)
    d = HipPdhgEngine.from_problem(p).layout_describe()
    for k in ("A", "At"):
        sw = d[k]["sweep"]
        assert d[k]["layout"] == "sweep" and sw["waves"] > 0 and sw["xcd_dealing"] in ("round robin", "contiguous eighths")
        assert sw["chosen_by"] in ("timing at create (tune_tiled_variant)", "static rule")
        if sw["chosen_by"].startswith("timing"):
            assert len(sw["candidates"]) >= 2 and all(c["ms"] > 0 for c in sw["candidates"])
    assert d["timing_at_create"] is True
    monkeypatch.setenv("PDHG_TUNE", "0")
    d0 = HipPdhgEngine.from_problem(p).layout_describe()
    assert d0["timing_at_create"] is False and all(d0[k]["sweep"]["chosen_by"] == "static rule" and not d0[k]["sweep"]["candidates"] for k in ("A", "At"))
