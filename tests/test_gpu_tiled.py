"""SpMV v2 (tiled-sweep layout) parity: forced on small problems with
PDHG_SPMV=tiled and tiny tiles so that many column tiles, run heads, padded
chunks, empty rows and long rows are all exercised.  Same bars as the stream
kernel: rows <= 2048 nnz bit-exact against the oracle."""
import os

import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (
    AdaptiveStepsizeParams, PdhgSolverState, take_step)
from oracle import oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[8, 10, 13, "cols=1000", "cols=97", "var=256", "var=4096"])
def tiled_env(request, monkeypatch):
    """Tile widths: powers of two (PDHG_TILE_SHIFT), arbitrary widths (PDHG_TILE_COLS), and
    equal-nonzero tiles of different widths (PDHG_VAR_TILES=1 on top of a nominal width).
    The value is the expected width of an entry's column field in bits (None: not fixed)."""
    monkeypatch.setenv("PDHG_SPMV", "tiled")
    if isinstance(request.param, str):
        kind, cols = request.param.split("=")
        monkeypatch.setenv("PDHG_TILE_COLS", cols)
        if kind == "var":
            monkeypatch.setenv("PDHG_VAR_TILES", "1")
            return None
        return (int(cols) - 1).bit_length()
    monkeypatch.setenv("PDHG_TILE_SHIFT", str(request.param))
    return request.param


@pytest.mark.parametrize("m,n,k,seed", [(1, 1, 1, 0), (7, 5, 3, 1), (300, 400, 10, 2),
                                        (5000, 3000, 10, 3), (20000, 30000, 10, 4),
                                        (1000, 50, 40, 5), (1025, 70000, 12, 6)])
def test_tiled_spmv_bit_exact(gpu_required, tiled_env, m, n, k, seed):
    p = random_lp(m, n, min(k, n), seed)
    A = p.constraint_matrix
    eng = HipPdhgEngine.from_problem(p)
    info = eng.layout_info()
    assert info["A_tiled_waves"] > 0 and info["At_tiled_waves"] > 0
    if tiled_env is not None:
        assert info["A_tile_shift"] == tiled_env
    elif n > int(os.environ["PDHG_TILE_COLS"]):
        assert info["var_tiles"] & 1
    rng = np.random.default_rng(seed)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    H.assert_products_match_oracle(eng, A, x, y, forced_sweep=True)      # bitwise for every row in strict order


def test_tiled_dense_block_runs(gpu_required, tiled_env):
    """Rows with many entries inside ONE tile: long same-row runs in a chunk."""
    import scipy.sparse as sp
    from firstorderlp_jl_amd import linear_programming_problem
    rng = np.random.default_rng(3)
    A = sp.csc_matrix(rng.standard_normal((70, 900)) * (rng.random((70, 900)) < 0.6))
    p = linear_programming_problem(np.zeros(900), np.ones(900), rng.standard_normal(900), 0.0,
                                   A, rng.standard_normal(70), 20)
    eng = HipPdhgEngine.from_problem(p)
    x, y = rng.standard_normal(900), rng.standard_normal(70)
    H.assert_products_match_oracle(eng, A, x, y, forced_sweep=True)


def test_tiled_with_long_and_empty_rows(gpu_required, tiled_env):
    p = H.skewed_lp(3000, 9000, seed=7, dense_rows=2, dense_cols=2)
    A = p.constraint_matrix
    m, n = A.shape
    eng = HipPdhgEngine.from_problem(p)
    info = eng.layout_info()
    assert info["A_long_rows"] == 2 and info["At_long_rows"] == 2 and info["A_tiled_waves"] > 0
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    ref = orc.spmv(m, n, A.indptr, A.indices, A.data, x)
    ref_t = orc.spmv_t(m, n, A.indptr, A.indices, A.data, y)
    got, got_t = eng.spmv(x), eng.spmv_t(y)
    short = np.diff(A.tocsr().indptr) <= 2048
    short_t = np.diff(A.indptr) <= 2048
    assert np.array_equal(got[short], ref[short]) and np.array_equal(got_t[short_t], ref_t[short_t])
    assert np.all(np.abs(got - ref) <= 1e-13 * (abs(A) @ np.abs(x)) + 1e-300)
    assert np.all(np.abs(got_t - ref_t) <= 1e-13 * (abs(A.T) @ np.abs(y)) + 1e-300)


def test_tiled_trajectory_matches_stream_and_oracle(gpu_required, tiled_env, monkeypatch):
    p = random_lp(4000, 5000, 10, 21)
    tiled = HipPdhgEngine.from_problem(p)
    monkeypatch.setenv("PDHG_SPMV", "stream")
    stream = HipPdhgEngine.from_problem(p)
    assert tiled.layout_info()["A_tiled_waves"] > 0 and stream.layout_info()["A_tiled_waves"] == 0
    st = H.oracle_from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    s1 = PdhgSolverState(tiled, step_size=step, primal_weight=pw)
    s2 = PdhgSolverState(stream, step_size=step, primal_weight=pw)
    st.step_size, st.primal_weight = step, pw
    for _ in range(40):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), s1)
        take_step(AdaptiveStepsizeParams(0.3, 0.6), s2)
        st.take_step_adaptive(0.3, 0.6)
    x1, y1 = tiled.get_current()
    x2, y2 = stream.get_current()
    # vectors are bit-exact per step in both layouts; only the block-partial
    # grouping of the reduction scalars differs between them
    np.testing.assert_allclose(x1, x2, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(y1, y2, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(x1, st.x, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(y1, st.y, rtol=1e-9, atol=1e-9)
    assert s1.total_number_iterations == st.total_number_iterations


def test_skewed_columns_get_equal_nonzero_tiles(gpu_required, monkeypatch):
    """A PageRank LP (Barabasi-Albert graph: the oldest nodes are hub columns AND hub rows):
    uniform tiles put most entries into the first tile; the layout notices (fullest tile
    > 1.5x average) and moves the boundaries.  Products stay bit-exact, the device rescaling
    walks the same tiles, and a trajectory matches the stream layout."""
    from firstorderlp_jl_amd.generators import pagerank_lp
    from firstorderlp_jl_amd.preprocess import rescale_problem
    p = pagerank_lp(60000, seed=4)
    A = p.constraint_matrix
    m, n = A.shape
    monkeypatch.setenv("PDHG_SPMV", "tiled")
    monkeypatch.setenv("PDHG_TILE_COLS", "2048")
    eng = HipPdhgEngine.from_problem(p)
    info = eng.layout_info()
    assert info["A_tiled_waves"] > 0 and info["var_tiles"] == 3
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    H.assert_products_match_oracle(eng, A, x, y, forced_sweep=True)
    # (engine against engine below: rows the sweep adds strictly in BOTH row orders -- at most 8 entries in relaxed order)
    limit = 2048 if os.environ.get("PDHG_ROW_ORDER") == "strict" else 8
    short = np.diff(A.tocsr().indptr) <= limit
    short_t = np.diff(A.indptr) <= limit
    # forcing uniform tiles gives the same products (short rows: bitwise)
    monkeypatch.setenv("PDHG_VAR_TILES", "0")
    uni = HipPdhgEngine.from_problem(p)
    assert uni.layout_info()["var_tiles"] == 0
    assert np.array_equal(uni.spmv(x)[short], eng.spmv(x)[short])
    monkeypatch.delenv("PDHG_VAR_TILES")
    # device rescaling over variable-width tiles == a fresh layout of the host-rescaled problem
    host = rescale_problem(10, False, None, 0, p)
    eng.rescale(10, False, None)
    ref = HipPdhgEngine.from_problem(host.scaled_qp)
    assert np.array_equal(eng.spmv(x)[short], ref.spmv(x)[short])
    assert np.array_equal(eng.spmv_t(y)[short_t], ref.spmv_t(y)[short_t])


def test_layout_construction_is_independent_of_host_threads(gpu_required, monkeypatch):
    """pdhg_create builds CSR(A) and both tiled layouts on host threads
    (PDHG_HOST_THREADS); 1 thread and 13 threads must give the same device
    layout: identical statistics and bitwise identical products (and both equal
    to the oracle's sequential loops)."""
    from firstorderlp_jl_amd.generators import random_lp
    from oracle import oracle as orc
    p = random_lp(700_000, 650_000, 7, seed=101)          # 4.9M nonzeros: above the threading threshold
    A = p.constraint_matrix
    rng = np.random.default_rng(5)
    x, y = rng.standard_normal(A.shape[1]), rng.standard_normal(A.shape[0])
    outs = []
    for threads in ("1", "13"):
        monkeypatch.setenv("PDHG_HOST_THREADS", threads)
        eng = HipPdhgEngine.from_problem(p)
        info = eng.layout_info()
        assert info["A_tiled_waves"] > 0 and info["At_tiled_waves"] > 0
        outs.append((info, eng.spmv(x), eng.spmv_t(y)))
        eng.close()
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    m, n = A.shape
    assert np.array_equal(outs[0][1], orc.spmv(m, n, A.indptr, A.indices, A.data, x))
    assert np.array_equal(outs[0][2], orc.spmv_t(m, n, A.indptr, A.indices, A.data, y))


def test_banded_matrix_keeps_the_stream_layout(gpu_required, monkeypatch):
    """Layout choice by locality: a matrix whose rows stay inside a narrow band of columns
    touches a few (workgroup, tile) cells only; its gathers are L2-local when streamed and
    the sweep would only add barriers (profiles/r02_locality.txt: 2x slower at 10M columns).
    The same shape with scattered columns gets the sweep.  Both layouts stay bit-exact."""
    import scipy.sparse as sp
    from firstorderlp_jl_amd import linear_programming_problem
    m = n = 600_000
    rng = np.random.default_rng(8)
    rows = np.repeat(np.arange(m), 4)

    def lp(cols):
        M = sp.csr_matrix((rng.standard_normal(4 * m), (rows, cols)), shape=(m, n))
        M.sum_duplicates()
        return linear_programming_problem(np.zeros(n), np.ones(n), rng.standard_normal(n), 0.0, M.tocsc(),
                                          rng.standard_normal(m), m // 2)

    banded = lp(np.clip(rows + rng.integers(-2000, 2001, 4 * m), 0, n - 1))
    scattered = lp(rng.integers(0, n, 4 * m))
    for k in ("PDHG_SPMV", "PDHG_TILE_SHIFT", "PDHG_TILE_COLS", "PDHG_VAR_TILES"):
        monkeypatch.delenv(k, raising=False)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    for p, want_tiled in ((banded, False), (scattered, True)):
        A = p.constraint_matrix
        eng = HipPdhgEngine.from_problem(p)
        info = eng.layout_info()
        assert (info["A_tiled_waves"] > 0) == want_tiled and (info["At_tiled_waves"] > 0) == want_tiled, info
        H.assert_products_match_oracle(eng, A, x, y, label="banded" if not want_tiled else "scattered")
        eng.close()
    monkeypatch.setenv("PDHG_SPMV", "tiled")          # forcing the sweep still works on the banded matrix
    A = banded.constraint_matrix
    eng = HipPdhgEngine.from_problem(banded)
    assert eng.layout_info()["A_tiled_waves"] > 0
    H.assert_products_match_oracle(eng, A, x, y, forced_sweep=True, label="banded, sweep forced")
    eng.close()


def test_layout_choice_long_rows_and_thin_cells(gpu_required, monkeypatch):
    """The sweep is chosen by what it needs, not by row length: the transposed side of a tall LP
    (rows of ~100 entries gathering from a long vector) is swept; a matrix with so few rows that
    a wave would get a handful of entries per tile streams.  Products bit-exact either way."""
    for k in ("PDHG_SPMV", "PDHG_TILE_SHIFT", "PDHG_TILE_COLS", "PDHG_VAR_TILES"):
        monkeypatch.delenv(k, raising=False)
    rng = np.random.default_rng(12)
    tall = random_lp(2_000_000, 400_000, 5, seed=31)          # A': 400K rows x 25 entries over 2M columns
    eng = HipPdhgEngine.from_problem(tall)
    info = eng.layout_info()
    assert info["At_tiled_waves"] > 0 and info["At_max_row_nnz"] > 25
    assert info["A_tiled_waves"] > 0                            # 400K columns = 3.2 MB: just beyond the 3 MiB threshold
    A = tall.constraint_matrix
    m, n = A.shape
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    assert np.array_equal(eng.spmv(x), orc.spmv(m, n, A.indptr, A.indices, A.data, x))
    assert np.array_equal(eng.spmv_t(y), orc.spmv_t(m, n, A.indptr, A.indices, A.data, y))
    eng.close()
    wide = random_lp(20_000, 8_000_000, 10, seed=32)           # 200K entries against a 64 MB vector: ~12 per (wave, tile) cell
    eng = HipPdhgEngine.from_problem(wide)
    info = eng.layout_info()
    assert info["A_tiled_waves"] == 0
    A = wide.constraint_matrix
    m, n = A.shape
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    assert np.array_equal(eng.spmv(x), orc.spmv(m, n, A.indptr, A.indices, A.data, x))
    assert np.array_equal(eng.spmv_t(y), orc.spmv_t(m, n, A.indptr, A.indices, A.data, y))
    eng.close()


def test_hub_rows_do_not_spill_into_an_extra_round(gpu_required, monkeypatch):
    """PageRank LP large enough that the per-wave entry cap makes more waves than one residency
    round holds (4096): the builder raises the rows per wave until they fit again.  Products of
    rows <= 2048 entries stay bit-exact with the widened waves."""
    from firstorderlp_jl_amd.generators import pagerank_lp
    p = pagerank_lp(400_000, seed=9)
    A = p.constraint_matrix
    m, n = A.shape
    monkeypatch.setenv("PDHG_SPMV", "tiled")
    monkeypatch.setenv("PDHG_TW_NNZ_CAP", "1.0")       # every hub region splits: well over 4096 waves at 98 rows per wave
    eng = HipPdhgEngine.from_problem(p)
    info = eng.layout_info()
    assert 0 < info["A_tiled_waves"] <= 4096 and 0 < info["At_tiled_waves"] <= 4096
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    H.assert_products_match_oracle(eng, A, x, y, forced_sweep=True)
    eng.close()


@pytest.mark.own_row_order
@pytest.mark.parametrize("order", ["strict", "relaxed"])
@pytest.mark.parametrize("variant", ["timed", "3", "4"])
@pytest.mark.parametrize("lo,hi", [(9, 12), (10, 30), (1, 20)], ids=["runs<=12", "runs<=30", "runs<=20"])
def test_runs_of_9_to_32_travel_lane_to_lane(gpu_required, monkeypatch, order, lo, hi, variant):
    """Rows of `lo` .. `hi` entries within +-300 columns of a random centre (the "clustered" shape of tools/shape_table.py):
    every row sits in one or two column tiles, so the sweep sees same-row runs of 9 .. 32 entries inside a tile and picks
    the lane-to-lane chunk variant (spmv_tiled_kernel<., 3>, csrc/spmv_kernels.hpp: tiled_chunk_scan) -- in BOTH row
    orders, because it adds a run in the sequential order: A x must equal the oracle's loops bit for bit on every row
    (the reference's `mul!` order, src/primal_dual_hybrid_gradient.jl:401-417 through SparseArrays).
    `variant`: pdhg_create times variant 3 against variant 4 (= 3, with variant 0's shuffle loop for chunks of short runs
    only) on the matrix and keeps the faster (tune_tiled_variant); PDHG_TW_MODE pins one.  Whichever runs, the same bits."""
    import scipy.sparse as sp
    from firstorderlp_jl_amd import linear_programming_problem, _lib
    monkeypatch.setenv("PDHG_ROW_ORDER", order)
    monkeypatch.setenv("PDHG_TILE_COLS", "4096")
    monkeypatch.setenv("PDHG_SPMV", "tiled")                # (the builder streams a matrix this small)
    if variant == "timed":
        monkeypatch.delenv("PDHG_TW_MODE", raising=False)
    else:
        monkeypatch.setenv("PDHG_TW_MODE", variant)
    rng = np.random.default_rng(lo * 100 + hi)
    m, n = 40_000, 30_000
    lens = rng.integers(lo, hi + 1, m)
    rows = np.repeat(np.arange(m), lens)
    cols = np.clip(np.repeat(rng.integers(0, n, m), lens) + rng.integers(-300, 301, rows.size), 0, n - 1)
    A = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(m, n))
    A.sum_duplicates()
    p = linear_programming_problem(np.zeros(n), np.full(n, 10.0), rng.standard_normal(n), 0.0, A.tocsc(),
                                   rng.standard_normal(m), m // 2)
    eng = HipPdhgEngine.from_problem(p)
    info = eng.layout_info()
    assert info["A_tiled_waves"] > 0, info
    want = (", 3>", ", 4>") if variant == "timed" else (f", {variant}>",)
    assert any(w in eng.kernel_name(_lib.K_SPMV_DUAL) for w in want), eng.kernel_name(_lib.K_SPMV_DUAL)
    Ac = p.constraint_matrix
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    assert np.array_equal(eng.spmv(x), orc.spmv(m, n, Ac.indptr, Ac.indices, Ac.data, x))
    seq_t = any(w in eng.kernel_name(_lib.K_SPMV_ATY) for w in (", 3>", ", 4>"))
    if info["At_tiled_waves"] > 0 and seq_t:
        assert np.array_equal(eng.spmv_t(y), orc.spmv_t(m, n, Ac.indptr, Ac.indices, Ac.data, y))
    else:
        H.assert_products_match_oracle(eng, Ac, x, y, forced_sweep=True)
    # and the fused products inside a trial step (A xbar with the dual step, A'y' with the interaction sums) from a
    # non-trivial iterate: the trial vectors bit for bit wherever the row sums are (tests/test_gpu_step_parity.py)
    o = H.oracle_from_problem(p)
    x0 = np.clip(rng.standard_normal(n), np.maximum(p.variable_lower_bound, -5), np.minimum(p.variable_upper_bound, 5))
    y0 = rng.standard_normal(m)
    y0[p.num_equalities:] = np.abs(y0[p.num_equalities:])
    eng.set_current(x0, y0)
    o.x, o.y = x0, y0
    o.recompute_dual_product()
    step, pw = H.initial_step_and_weight(p)
    both_exact = order == "strict" or seq_t
    for theta in (1.0, 0.37):
        raw = eng.trial_step(step, pw, theta)
        raw_o, xn, yn, an = o.trial_step(step, pw, theta)
        gx, gy, ga = eng.get_trial()
        if both_exact:
            assert np.array_equal(gy, yn) and np.array_equal(gx, xn) and np.array_equal(ga, an)
        else:                                                          # (x' comes from A'y, summed by the relaxed rules)
            np.testing.assert_allclose(gy, yn, rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(gx, xn, rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(ga, an, rtol=1e-11, atol=1e-11)
        scale = np.array([np.abs(raw_o[1] * raw_o[2]) ** 0.5 + abs(raw_o[0]), raw_o[1], raw_o[2]])
        assert np.all(np.abs(raw[:3] - raw_o[:3]) <= 1e-12 * scale + 1e-300)
    eng.close()


def test_row_groups_dealt_to_the_xcds_in_contiguous_eighths(gpu_required, monkeypatch):
    """spmv_tiled_kernel's XCD remap (per_xcd > 0: workgroup b takes row group (b % 8) * per_xcd + b / 8; the builder turns
    it on for banded sweeps, PDHG_TW_REMAP forces it): the same row groups, the same block-partial slots -- products,
    trial vectors AND the trial's scalars must be bit for bit those of the plain numbering, on a grid that is not a
    multiple of 8 (idle workgroups at the end of the last XCD's share)."""
    monkeypatch.setenv("PDHG_SPMV", "tiled")
    monkeypatch.setenv("PDHG_TILE_COLS", "4096")
    p = random_lp(21_000, 30_000, 10, seed=77)
    A = p.constraint_matrix
    m, n = A.shape
    rng = np.random.default_rng(5)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    x0 = np.clip(rng.standard_normal(n), np.maximum(p.variable_lower_bound, -5), np.minimum(p.variable_upper_bound, 5))
    y0 = rng.standard_normal(m)
    y0[p.num_equalities:] = np.abs(y0[p.num_equalities:])
    step, pw = H.initial_step_and_weight(p)
    out = []
    for remap in ("0", "1"):
        monkeypatch.setenv("PDHG_TW_REMAP", remap)
        for graph in ("0", "1"):
            monkeypatch.setenv("PDHG_GRAPH", graph)
            monkeypatch.setenv("PDHG_GRAPH_TILED", "1")
            eng = HipPdhgEngine.from_problem(p)
            info = eng.layout_info()
            assert info["A_tiled_waves"] >= 16 * 8 and info["A_tiled_waves"] % 64 != 0, info
            H.assert_products_match_oracle(eng, A, x, y, forced_sweep=True)
            eng.set_current(x0, y0)
            raw = [np.array(eng.trial_step(step, pw, theta)) for theta in (1.0, 0.37)]
            out.append((eng.spmv(x), eng.spmv_t(y), raw, eng.get_trial()))
            eng.close()
    for o in out[1:]:
        assert np.array_equal(o[0], out[0][0]) and np.array_equal(o[1], out[0][1])
        assert all(np.array_equal(a, b) for a, b in zip(o[2], out[0][2]))
        assert all(np.array_equal(a, b) for a, b in zip(o[3], out[0][3]))


def test_timed_sweep_policies_cannot_change_a_bit(gpu_required, monkeypatch):
    """pdhg_create times the sweep's chunk variant (3 against 4) and the dealing of row groups to the XCDs on the matrix
    (tune_tiled_variant).  Whatever wins, products, trial vectors and the trial's scalars are those of the untimed
    choice: PDHG_TW_TUNE=0 against the default, on a matrix whose runs put it on variant 3."""
    import scipy.sparse as sp
    from firstorderlp_jl_amd import linear_programming_problem
    monkeypatch.setenv("PDHG_SPMV", "tiled")
    monkeypatch.setenv("PDHG_TILE_COLS", "4096")
    monkeypatch.delenv("PDHG_TW_MODE", raising=False)
    monkeypatch.delenv("PDHG_TW_REMAP", raising=False)
    rng = np.random.default_rng(31)
    m, n = 50_000, 30_000
    lens = np.clip(np.exp(rng.normal(1.8, 1.0, m)).astype(np.int64), 1, 25)
    rows = np.repeat(np.arange(m), lens)
    cols = np.clip(np.repeat(rng.integers(0, n, m), lens) + rng.integers(-40, 41, rows.size), 0, n - 1)
    A = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(m, n))
    A.sum_duplicates()
    p = linear_programming_problem(np.zeros(n), np.full(n, 10.0), rng.standard_normal(n), 0.0, A.tocsc(),
                                   rng.standard_normal(m), m // 2)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    x0 = np.clip(rng.standard_normal(n), 0.0, 5.0)
    y0 = rng.standard_normal(m)
    y0[p.num_equalities:] = np.abs(y0[p.num_equalities:])
    step, pw = H.initial_step_and_weight(p)
    out = []
    for tune in ("0", "1"):
        monkeypatch.setenv("PDHG_TW_TUNE", tune)
        eng = HipPdhgEngine.from_problem(p)
        assert eng.layout_info()["A_tiled_waves"] > 0
        eng.set_current(x0, y0)
        raw = [np.array(eng.trial_step(step, pw, theta)) for theta in (1.0, 0.37)]
        out.append((eng.spmv(x), eng.spmv_t(y), raw, eng.get_trial()))
        eng.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert all(np.array_equal(a, b) for a, b in zip(out[0][2], out[1][2]))
    assert all(np.array_equal(a, b) for a, b in zip(out[0][3], out[1][3]))
