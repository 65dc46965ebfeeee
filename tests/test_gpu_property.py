"""Property test (hypothesis): random small LPs -- shapes from 0 x 1 up, densities from
empty to dense, duplicated structure, +-Inf bounds, any number of equalities -- through
every way the library can lay the matrix out (stream, tiled with tiny tiles, column-slab
passes, 2- and 3-shard groups, with and without the one-graph-launch trial).  For each:
the products A x and A'y, the vectors of a trial step (x', y', A'y'), of a second trial after an
accept, and the running average must be
BIT-IDENTICAL to the CPU oracle (rows are far below the long-row threshold), and the
five step scalars must agree to the condition-aware bound 1e-13 * sum|terms|."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from firstorderlp_jl_amd import HipPdhgEngine, linear_programming_problem
from oracle import oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu

LAYOUTS = ["default", "stream_nograph", "tiled6", "tiled8", "tiled_cols96", "tiled_var", "slabs", "shards2", "shards3",
           "shards2_tiled"]


@st.composite
def small_lps(draw):
    m = draw(st.integers(0, 70))
    n = draw(st.integers(1, 200))
    density = draw(st.sampled_from([0.0, 0.03, 0.15, 0.5, 1.0]))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    A = sp.random(m, n, density=density, format="csc", random_state=np.random.RandomState(seed % (2 ** 31)),
                  data_rvs=rng.standard_normal)
    A.sort_indices()
    lb = np.where(rng.random(n) < 0.25, -np.inf, rng.integers(-2, 1, n).astype(float))
    ub = np.where(rng.random(n) < 0.4, np.inf, np.maximum(lb, 0.0) + rng.integers(0, 3, n))
    c = rng.standard_normal(n) * (rng.random(n) < 0.8)
    b = rng.standard_normal(m)
    num_eq = draw(st.integers(0, m))
    step = float(draw(st.sampled_from([1e-3, 0.1, 1.0, 7.5])))
    pw = float(draw(st.sampled_from([0.01, 1.0, 30.0])))
    theta = float(draw(st.sampled_from([1.0, 0.37])))
    return linear_programming_problem(lb, ub, c, 0.0, A, b, num_eq), seed, step, pw, theta


def _engine(p, layout):
    keys = ("PDHG_SPMV", "PDHG_TILE_SHIFT", "PDHG_TILE_COLS", "PDHG_VAR_TILES", "PDHG_GRAPH", "PDHG_SLABS", "PDHG_SLAB_MB")
    saved = {k: os.environ.pop(k, None) for k in keys}
    kw = {}
    try:
        if layout == "stream_nograph":
            os.environ.update(PDHG_SPMV="stream", PDHG_GRAPH="0")
        elif layout == "tiled_cols96":       # a width that is not a power of two
            os.environ.update(PDHG_SPMV="tiled", PDHG_TILE_COLS="96")
        elif layout == "tiled_var":          # equal-nonzero tiles of different widths
            os.environ.update(PDHG_SPMV="tiled", PDHG_TILE_COLS="64", PDHG_VAR_TILES="1")
        elif layout.startswith("tiled"):
            os.environ.update(PDHG_SPMV="tiled", PDHG_TILE_SHIFT=layout[5:])
        elif layout == "slabs":
            os.environ.update(PDHG_SPMV="stream", PDHG_SLABS="1", PDHG_SLAB_MB="0.00005")
        elif layout == "shards2_tiled":
            os.environ.update(PDHG_SPMV="tiled", PDHG_TILE_SHIFT="6")
            kw["device_ids"] = [0, 0]
        elif layout.startswith("shards"):
            kw["device_ids"] = [0] * int(layout[6:])
        return HipPdhgEngine.from_problem(p, **kw)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


@settings(max_examples=200, deadline=None, suppress_health_check=list(HealthCheck))
@given(case=small_lps(), layout=st.sampled_from(LAYOUTS))
def test_products_and_one_trial_are_bit_identical_to_the_oracle(gpu_required, case, layout):
    p, seed, step, pw, theta = case
    # A sweep FORCED onto a small dense-ish matrix with 64-column tiles has same-row runs of dozens of entries inside a
    # tile; in the shipped relaxed order such chunks are tree-reduced (tiled_chunk_relaxed), so bit-identity with the
    # sequential sums is a strict-order property there (the builder itself would stream these matrices).  The relaxed
    # bound for forced sweeps is checked in test_gpu_tiled.py / test_gpu_row_order.py.
    from hypothesis import assume
    assume(not (os.environ.get("PDHG_ROW_ORDER") == "relaxed" and "tiled" in layout))
    A = p.constraint_matrix
    m, n = A.shape
    eng = _engine(p, layout)
    oracle = H.oracle_from_problem(p)
    try:
        rng = np.random.default_rng(seed + 1)
        x, y = rng.standard_normal(n), rng.standard_normal(m)
        # start from a generic point so that x - tau*g, the projections and A'y all matter
        eng.set_current(x, y)
        oracle.x, oracle.y = x.copy(), y.copy()
        oracle.recompute_dual_product()
        sharded = layout.startswith("shards")
        got_aty = eng.get_dual_product()
        if sharded:       # A'y is a sum of per-shard partials: rank-ordered, not column-ordered
            np.testing.assert_allclose(got_aty, oracle.aty, rtol=0, atol=1e-13 * (abs(A.T) @ np.abs(y)).max(initial=0) + 1e-300)
            oracle.aty = got_aty.copy()       # continue from the same bits
        else:
            assert np.array_equal(got_aty, oracle.aty)
        assert np.array_equal(eng.spmv(x), orc.spmv(m, n, A.indptr, A.indices, A.data, x)), layout
        raw = eng.trial_step(step, pw, theta)
        raw_o, xn, yn, an = oracle.trial_step(step, pw, theta)
        gx, gy, ga = eng.get_trial()
        assert np.array_equal(gx, xn), layout
        assert np.array_equal(gy, yn), layout
        if sharded:
            np.testing.assert_allclose(ga, an, rtol=0, atol=1e-13 * (abs(A.T) @ np.abs(yn)).max(initial=0) + 1e-300)
        else:
            assert np.array_equal(ga, an), layout
        dx, dd, dy = xn - oracle.x, ga - oracle.aty, yn - oracle.y
        bounds = [np.sum(np.abs(dx * dd)), np.sum(dx * dx), np.sum(dy * dy), np.sum(dd * dd)]
        exact = [float(dx @ dd), float(dx @ dx), float(dy @ dy), float(dd @ dd)]
        for q in range(4):
            assert abs(raw[q] - exact[q]) <= 1e-13 * bounds[q] + 1e-300, (layout, q, raw[q], exact[q])
        assert raw[4] == 0.0
        # accept (the running sums are updated lazily, by the next trial's kernels), a second
        # trial from the accepted point, then the average
        oracle.step_size = step
        eng.accept(step)
        oracle.accept(gx, gy, ga)
        eng.trial_step(0.7 * step, pw, theta)
        _, xn2, yn2, an2 = oracle.trial_step(0.7 * step, pw, theta)
        gx2, gy2, ga2 = eng.get_trial()
        assert np.array_equal(gx2, xn2), layout
        assert np.array_equal(gy2, yn2), layout
        if sharded:
            np.testing.assert_allclose(ga2, an2, rtol=0, atol=1e-13 * (abs(A.T) @ np.abs(yn2)).max(initial=0) + 1e-300)
        else:
            assert np.array_equal(ga2, an2), layout
        xa, ya = eng.get_average()
        xo, yo = oracle.compute_average()
        assert np.array_equal(xa, xo) and np.array_equal(ya, yo), layout
    finally:
        eng.close()
        oracle.close()
