"""The CSR row blocks as a persistent software-pipelined launch (spmv_stream_pipe_kernel, csrc/spmv_kernels.hpp): the same
lane-per-row sums in the same order as spmv_stream_kernel, so NOT A BIT may differ from it -- products, fused epilogues,
block partials (hence accept / reject decisions), in both row orders, with column-slab passes (INIT carry), long rows,
blocks of more than 256 very short rows (whose later row trips are not pre-requested), empty rows, as graph nodes and as
plain launches.  The builder picks it for stream-class matrices with more than 1 024 row blocks that did not get the
sliced jagged copy (csrc/sj_kernels.hpp), and -- round 6 -- for every matrix of 16 ... 1 024 row blocks (one workgroup per block:
its request order alone)."""
import numpy as np
import pytest
import scipy.sparse as sp

from firstorderlp_jl_amd import HipPdhgEngine, linear_programming_problem
from firstorderlp_jl_amd.generators import pagerank_lp, random_lp
from tests import helpers as H
from tests.test_gpu_sj import _ragged_lp, _run

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _engine(p, monkeypatch, pipe, graph="1", slab_mb=None):
    monkeypatch.setenv("PDHG_SPMV", "stream")
    monkeypatch.setenv("PDHG_COOP", "0")
    monkeypatch.setenv("PDHG_GRAPH", graph)
    monkeypatch.setenv("PDHG_SJ", "0")
    monkeypatch.setenv("PDHG_STREAM_PIPE", pipe)
    if slab_mb is None:
        monkeypatch.setenv("PDHG_SLABS", "0")
    else:
        monkeypatch.setenv("PDHG_SLABS", "1")
        monkeypatch.setenv("PDHG_SLAB_MB", str(slab_mb))
    return HipPdhgEngine.from_problem(p)


def _very_short_rows(m, n, seed):
    """1-3 entries per row: row blocks of 1 024 rows, i.e. four row trips per block."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 4, m)
    rows = np.repeat(np.arange(m), lens)
    A = sp.csr_matrix((rng.standard_normal(rows.size), (rows, rng.integers(0, n, rows.size))), shape=(m, n)).tocsc()
    A.sum_duplicates()
    A.sort_indices()
    return linear_programming_problem(np.zeros(n), np.full(n, 5.0), rng.standard_normal(n), 0.0, A, rng.standard_normal(m), m // 3)


@pytest.mark.parametrize("maker", [lambda: random_lp(30_000, 20_000, 6, seed=21), lambda: _ragged_lp(5_000, 7_001, seed=2),
                                   lambda: H.skewed_lp(3_000, 9_000, seed=7, dense_rows=2, dense_cols=2),
                                   lambda: _very_short_rows(40_000, 30_000, seed=5), lambda: random_lp(70, 50, 3, seed=1)],
                         ids=["random", "ragged", "skewed_long_rows", "very_short_rows", "tiny"])
def test_pipelined_launch_is_bitwise_the_plain_kernel(gpu_required, monkeypatch, maker):
    p = maker()
    A = p.constraint_matrix
    m, n = A.shape
    eng = _engine(p, monkeypatch, "1")
    info = eng.layout_info()
    assert info["A_pipe"] == 1 and info["At_pipe"] == 1 and info["A_sj"] == 0, info
    assert "spmv_stream_pipe_kernel" in eng.kernel_name(1) and "spmv_stream_pipe_kernel" in eng.kernel_name(2)
    plain = _engine(p, monkeypatch, "0")
    assert plain.layout_info()["A_pipe"] == 0
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    assert np.array_equal(eng.spmv(x), plain.spmv(x)) and np.array_equal(eng.spmv_t(y), plain.spmv_t(y))
    H.assert_products_match_oracle(eng, A, x, y, label="pipelined")
    r_pipe = _run(eng, p)
    r_pipe_plain_launches = _run(_engine(p, monkeypatch, "1", graph="0"), p)
    r_plain = _run(plain, p)
    for a, b, c in zip(r_pipe, r_pipe_plain_launches, r_plain):
        assert np.array_equal(a, b) and np.array_equal(a, c)


@pytest.mark.parametrize("maker,slab_mb", [(lambda: random_lp(200_000, 150_000, 8, seed=3), 0.5),
                                           (lambda: pagerank_lp(120_000, seed=4), 0.3)], ids=["random", "pagerank"])
def test_pipelined_slab_passes_are_bitwise_the_plain_kernel(gpu_required, monkeypatch, maker, slab_mb):
    p = maker()
    eng = _engine(p, monkeypatch, "1", slab_mb=slab_mb)
    info = eng.layout_info()
    assert 2 <= info["A_slabs"] <= 4 and info["A_pipe"] == 1 and info["At_pipe"] == 1, info
    r_pipe = _run(eng, p)
    r_plain = _run(_engine(p, monkeypatch, "0", slab_mb=slab_mb), p)
    for a, c in zip(r_pipe, r_plain):
        assert np.array_equal(a, c)


def test_the_builder_picks_the_pipelined_launch(gpu_required, monkeypatch):
    """Rows of 1 to 40 entries in no order (a 256-row group is a third full against its longest row: not the sliced jagged
    layout), > 1 024 row blocks, stream class: the pipelined launch of the row blocks.  A PageRank LP (hub rows: blocks of
    very different cost) keeps the plain kernel's dynamic dispatch."""
    monkeypatch.delenv("PDHG_SJ", raising=False)
    monkeypatch.delenv("PDHG_STREAM_PIPE", raising=False)
    monkeypatch.setenv("PDHG_SPMV", "stream")
    m, n = 400_000, 300_000
    rng = np.random.default_rng(6)
    lens = np.minimum(rng.geometric(0.12, m), 40)
    rows = np.repeat(np.arange(m), lens)
    M = sp.csr_matrix((rng.standard_normal(rows.size), (rows, rng.integers(0, n, rows.size))), shape=(m, n))
    M.sum_duplicates()
    p = linear_programming_problem(np.zeros(n), np.full(n, 10.0), rng.standard_normal(n), 0.0, M.tocsc(), rng.standard_normal(m), m // 2)
    info = HipPdhgEngine.from_problem(p).layout_info()
    assert info["A_tiled_waves"] == 0 and info["A_blocks"] > 1024 and info["A_sj"] == 0 and info["A_pipe"] == 1, info
    monkeypatch.delenv("PDHG_SPMV", raising=False)
    hub = HipPdhgEngine.from_problem(pagerank_lp(330_000, seed=4)).layout_info()
    assert hub["A_blocks"] > 1024 and hub["A_pipe"] == 0 and hub["A_sj"] == 0, hub
    # round 6: at most 1 024 row blocks -> one workgroup per block in the pipelined kernel's request order (one round trip for
    # entries, row extents and operands), whatever the rows; a handful of blocks: the plain kernel
    small = HipPdhgEngine.from_problem(random_lp(5000, 4000, 8, seed=7)).layout_info()
    assert 16 <= small["A_blocks"] <= 1024 and small["A_pipe"] == 1 and small["At_pipe"] == 1, small
    tiny = HipPdhgEngine.from_problem(random_lp(700, 500, 6, seed=7)).layout_info()
    assert tiny["A_blocks"] < 16 and tiny["A_pipe"] == 0, tiny
