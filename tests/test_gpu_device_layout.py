"""Layout construction on the device (csrc/device_layout.hpp): the caller's CSC arrays are
uploaded as they are, CSR(A') is a narrowing, CSR(A) a stable radix sort by row, the sweep's
tile-major copy a stable counting sort per wave; only row pointers and per-(wave, tile) counts
travel back for the host-side planning.  The result must be BIT-IDENTICAL to the host
builders': every device array of both layouts is compared through order-sensitive checksums
(pdhg_layout_checksums), and the products against the CPU oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

from firstorderlp_jl_amd import HipPdhgEngine, linear_programming_problem
from firstorderlp_jl_amd.generators import pagerank_lp, random_lp
from oracle import oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _both(p, monkeypatch, **env):
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PDHG_DEVICE_LAYOUT", mode)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = HipPdhgEngine.from_problem(p)
        out[mode] = (eng.layout_checksums(), eng.layout_info(), eng)
    return out


def _with_structure(m, n, k, seed, dense_rows=0, dense_cols=0, empty_rows=0, empty_cols=0):
    """random_lp plus rows / columns of thousands of entries (long rows in either layout) and empty rows / columns."""
    rng = np.random.default_rng(seed)
    p = random_lp(m, n, k, seed)
    A = p.constraint_matrix.tocsr()
    parts = []
    for r in range(dense_rows):
        cols = np.sort(rng.choice(n, min(n, 5000 + 900 * r), replace=False))
        parts.append(sp.csr_matrix((rng.standard_normal(len(cols)), (np.zeros(len(cols), dtype=int), cols)), shape=(1, n)))
    if parts:
        A = sp.vstack(parts + [A[dense_rows:]]).tocsr()
    A = A.tocsc()
    parts = []
    for c in range(dense_cols):
        rows = np.sort(rng.choice(m, min(m, 4000 + 700 * c), replace=False))
        parts.append(sp.csc_matrix((rng.standard_normal(len(rows)), (rows, np.zeros(len(rows), dtype=int))), shape=(m, 1)))
    if parts:
        A = sp.hstack(parts + [A[:, dense_cols:]]).tocsc()
    if empty_rows or empty_cols:
        row_keep = np.ones(m)
        row_keep[100:100 + 7 * empty_rows:7] = 0.0
        col_keep = np.ones(n)
        col_keep[200:200 + empty_cols] = 0.0
        A = (sp.diags(row_keep) @ A @ sp.diags(col_keep)).tocsc()
        A.eliminate_zeros()
    A.sort_indices()
    return linear_programming_problem(p.variable_lower_bound, p.variable_upper_bound, p.objective_vector, 0.0,
                                      A, p.right_hand_side, p.num_equalities)


CASES = {
    "tiled_both": (lambda: random_lp(700_000, 650_000, 7, seed=101), {}),
    "tiled_small_tiles": (lambda: random_lp(90_000, 80_000, 9, seed=5), {"PDHG_SPMV": "tiled", "PDHG_TILE_COLS": "4096"}),
    "stream_small": (lambda: random_lp(30_000, 20_000, 6, seed=21), {}),
    "long_rows_and_empties": (lambda: _with_structure(60_000, 50_000, 8, seed=9, dense_rows=3, dense_cols=2,
                                                      empty_rows=5, empty_cols=40), {}),
    "long_rows_forced_sweep": (lambda: _with_structure(60_000, 50_000, 8, seed=9, dense_rows=3, dense_cols=2),
                               {"PDHG_SPMV": "tiled", "PDHG_TILE_COLS": "8192"}),
    "pagerank_slabs": (lambda: pagerank_lp(150_000, seed=4), {"PDHG_SLAB_MB": "0.4"}),
    "pagerank_var_tiles": (lambda: pagerank_lp(150_000, seed=4), {"PDHG_SPMV": "tiled", "PDHG_TILE_COLS": "8192"}),
    "var_tiles_forced": (lambda: random_lp(90_000, 80_000, 9, seed=6), {"PDHG_SPMV": "tiled", "PDHG_TILE_COLS": "4096",
                                                                        "PDHG_VAR_TILES": "1"}),
    "wide": (lambda: random_lp(40_000, 900_000, 12, seed=3), {}),
    "tall": (lambda: random_lp(900_000, 40_000, 5, seed=3), {}),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_device_built_layouts_are_bit_identical_to_the_host_builders(gpu_required, monkeypatch, name):
    maker, env = CASES[name]
    p = maker()
    got = _both(p, monkeypatch, **env)
    (ck_h, info_h, eng_h), (ck_d, info_d, eng_d) = got["0"], got["1"]
    assert info_h == info_d, (info_h, info_d)
    if name.endswith("var_tiles") or name.startswith("var_tiles"):
        assert info_d["var_tiles"] != 0 and info_d["A_tiled_waves"] > 0          # tiles of different widths, built on the device
    if name == "pagerank_slabs":
        assert info_d["A_slabs"] >= 2 and info_d["At_slabs"] >= 2      # their arrays ride in the sweep's checksum slots
    labels = ["rowptr", "col", "val", "blks", "long_row", "long_chunk_ptr", "chunk_row", "chunk_off", "pk", "tv",
              "wave_rows", "wave_ent", "wave_step_off", "step_tile", "wg_step_off", "plan"]
    bad = [("A" if q < 16 else "At") + "." + labels[q % 16] for q in range(32) if ck_h[q] != ck_d[q]]
    assert not bad, bad
    A = p.constraint_matrix
    m, n = A.shape
    rng = np.random.default_rng(2)
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    assert np.array_equal(eng_d.spmv(x), eng_h.spmv(x)) and np.array_equal(eng_d.spmv_t(y), eng_h.spmv_t(y))
    H.assert_products_match_oracle(eng_d, A, x, y, forced_sweep=env.get("PDHG_SPMV") == "tiled", label=name)
    eng_h.close()
    eng_d.close()


def test_device_ingest_validates_like_the_host_path(gpu_required, monkeypatch):
    monkeypatch.setenv("PDHG_DEVICE_LAYOUT", "1")
    p = random_lp(2000, 1500, 6, seed=1)
    A = p.constraint_matrix.copy()
    bad = A.copy()
    bad.indices = bad.indices.copy()
    bad.indices[17] = A.shape[0] + 3                           # row index out of range
    with pytest.raises(Exception, match="rowval out of range"):
        HipPdhgEngine(bad, p.objective_vector, p.right_hand_side, p.variable_lower_bound, p.variable_upper_bound, 0)
    bad = A.copy()
    bad.indptr = bad.indptr.copy()
    bad.indptr[40] = bad.indptr[39] - 1 if bad.indptr[39] > 0 else bad.indptr[41] + 1      # not monotone
    with pytest.raises(Exception, match="colptr"):
        HipPdhgEngine(bad, p.objective_vector, p.right_hand_side, p.variable_lower_bound, p.variable_upper_bound, 0)


def test_device_layout_trajectory_and_shards(gpu_required, monkeypatch):
    """A whole adaptive run and a two-shard group (every shard's slice goes through the same ingest)."""
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_step
    p = random_lp(120_000, 100_000, 8, seed=11)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PDHG_DEVICE_LAYOUT", mode)
        for ids in (None, [0, 0]):
            eng = HipPdhgEngine.from_problem(p) if ids is None else HipPdhgEngine.from_problem(p, device_ids=ids)
            step, pw = H.initial_step_and_weight(p)
            st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
            for _ in range(40):
                take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
            outs[(mode, ids is None)] = (np.concatenate(eng.get_current()), st.total_number_iterations, st.step_size)
            eng.close()
    for single in (True, False):
        a, b = outs[("0", single)], outs[("1", single)]
        assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]
