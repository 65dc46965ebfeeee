"""N2 on the device: pdhg_rescale against the host rescale_problem
(firstorderlp.jl_amd/preprocess.py, itself pinned by the reference's
test_qp_processing.jl KATs).  Ruiz (max-based) is bit-exact: same per-entry
multiplication order; L2 / Pock-Chambolle factors come from sums whose order
differs (wave tree vs sequential) -> 1e-13 relative."""
import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import pagerank_lp, random_lp
from firstorderlp_jl_amd.preprocess import rescale_problem
from tests import helpers as H

pytestmark = pytest.mark.gpu

MAKERS = {"random": lambda: random_lp(3000, 4000, 8, 3),
          "skewed": lambda: H.skewed_lp(2500, 6000, 5),
          "pagerank": lambda: pagerank_lp(20000, seed=2),
          "example_lp": H.example_lp}


@pytest.mark.parametrize("name", sorted(MAKERS))
@pytest.mark.parametrize("ruiz,l2,alpha", [(10, False, None), (3, False, 1.0), (0, True, None),
                                           (2, True, 0.0), (0, False, 2.0), (0, False, None)])
def test_device_rescale_matches_host(gpu_required, name, ruiz, l2, alpha):
    p = MAKERS[name]()
    host = rescale_problem(ruiz, l2, alpha, 0, p)
    eng = HipPdhgEngine.from_problem(p)
    E, D = eng.rescale(ruiz, l2, alpha)
    c, b, lb, ub = eng.get_problem_vectors()
    exact = not l2 and alpha is None
    def close(a, w):
        if exact:
            assert np.array_equal(a, w)
        else:
            np.testing.assert_allclose(a, w, rtol=1e-12, atol=0)
    close(E, host.constraint_rescaling)
    close(D, host.variable_rescaling)
    s = host.scaled_qp
    close(c, s.objective_vector); close(b, s.right_hand_side)
    close(lb, s.variable_lower_bound); close(ub, s.variable_upper_bound)
    mx = eng.matrix_max_abs()
    want = float(np.abs(s.constraint_matrix.data).max())
    assert mx == want if exact else abs(mx - want) <= 1e-12 * want
    # the rescaled resident matrix itself, through both layouts
    ref = HipPdhgEngine.from_problem(s)
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(eng.n), rng.standard_normal(eng.m)
    for got, wantv in ((eng.spmv(x), ref.spmv(x)), (eng.spmv_t(y), ref.spmv_t(y))):
        if exact:
            assert np.array_equal(got, wantv)
        else:
            np.testing.assert_allclose(got, wantv, rtol=1e-11, atol=1e-11 * np.abs(wantv).max())


@pytest.mark.parametrize("tile_env", [("PDHG_TILE_SHIFT", "9"), ("PDHG_TILE_COLS", "700")])
def test_device_rescale_tiled_layout(gpu_required, monkeypatch, tile_env):
    monkeypatch.setenv("PDHG_SPMV", "tiled")
    monkeypatch.setenv(*tile_env)
    p = random_lp(5000, 7000, 9, 11)
    host = rescale_problem(10, False, None, 0, p)
    eng = HipPdhgEngine.from_problem(p)
    assert eng.layout_info()["A_tiled_waves"] > 0
    eng.rescale(10, False, None)
    ref = HipPdhgEngine.from_problem(host.scaled_qp)
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(eng.n), rng.standard_normal(eng.m)
    assert np.array_equal(eng.spmv(x), ref.spmv(x))
    assert np.array_equal(eng.spmv_t(y), ref.spmv_t(y))


@pytest.mark.parametrize("ruiz,l2,alpha", [(10, False, None), (4, False, 1.0), (0, True, None)])
def test_device_rescale_qp(gpu_required, ruiz, l2, alpha):
    """QP: the Ruiz column factors take the max over [A; Q] columns and the
    objective matrix is scaled to (D^-1 Q) D^-1 on the device (preprocess.jl:425-433,
    562-564).  Checked through the scaling vectors and through a trial step, whose
    primal update multiplies by the resident Q."""
    import scipy.sparse as sp
    from firstorderlp_jl_amd.generators import random_lp
    p = random_lp(1500, 1200, 6, seed=13)
    B = sp.random(1200, 1200, density=0.004, random_state=9, format="csc")
    p.objective_matrix = sp.csc_matrix(5.0 * (B.T @ B) + sp.diags(np.random.default_rng(1).uniform(0.0, 3.0, 1200)))
    host = rescale_problem(ruiz, l2, alpha, 0, p)
    eng = HipPdhgEngine.from_problem(p)
    E, D = eng.rescale(ruiz, l2, alpha)
    exact = not l2 and alpha is None
    def close(a, w):
        if exact:
            assert np.array_equal(a, w)
        else:
            np.testing.assert_allclose(a, w, rtol=1e-12, atol=0)
    close(E, host.constraint_rescaling)
    close(D, host.variable_rescaling)
    ref = HipPdhgEngine.from_problem(host.scaled_qp)
    rng = np.random.default_rng(2)
    x0, y0 = np.abs(rng.standard_normal(eng.n)), rng.standard_normal(eng.m)
    for e in (eng, ref):
        e.set_current(x0, y0)
    ra, rb = eng.trial_step(0.05, 1.3), ref.trial_step(0.05, 1.3)
    ta, tb = eng.get_trial(), ref.get_trial()
    if exact:
        assert np.array_equal(ra, rb)
        assert all(np.array_equal(u, v) for u, v in zip(ta, tb))
    else:
        np.testing.assert_allclose(ra, rb, rtol=1e-9)
        for u, v in zip(ta, tb):
            np.testing.assert_allclose(u, v, rtol=1e-10, atol=1e-12)


# ---- the reference's own exact-value cases (test/test_qp_processing.jl) through pdhg_rescale ----
def _ref_lp(A, b, lb=(0.0, 0.0)):
    from firstorderlp_jl_amd import linear_programming_problem
    return linear_programming_problem(list(lb), [1.0, 2.0], [1.0, 2.0], 0.0, A, b, 1)


S = np.sqrt
REFERENCE_CASES = {
    # name: (A, rhs, lb, (ruiz, l2, alpha), constraint_rescaling, variable_rescaling)
    "l2 :234": ([[1.0, 1.0], [1.0, -1.0], [1.0, 0.0]], [1.0, 1.0, 2.0], (0.0, 0.0), (0, True, None),
                [2 ** 0.25, 2 ** 0.25, 1.0], [3 ** 0.25, 2 ** 0.25]),
    "l2 empty row :269": ([[1.0, 1.0], [1.0, -1.0], [0.0, 0.0]], [1.0, 1.0, 0.0], (0.0, 0.0), (0, True, None),
                          [2 ** 0.25, 2 ** 0.25, 1.0], [2 ** 0.25, 2 ** 0.25]),
    "l2 empty column :304": ([[1.0, 0.0], [1.0, 0.0], [2.0, 0.0]], [1.0, 1.0, 2.0], (0.0, 0.0), (0, True, None),
                             [1.0, 1.0, S(2)], [6 ** 0.25, 1.0]),
    "pock-chambolle 0 :339": ([[1.0, 1.0], [2.0, -1.0], [1.0, 0.0]], [1.0, 1.0, 2.0], (-1.0, -1.0), (0, False, 0.0),
                              [S(2), S(2), S(2)], [S(6), S(2)]),
    "pock-chambolle 1 :359": ([[1.0, 1.0], [2.0, -1.0], [1.0, 0.0]], [1.0, 1.0, 2.0], (-1.0, -1.0), (0, False, 1.0),
                              [S(2), S(3), S(1)], [S(4), S(2)]),
    "pock-chambolle 2 :379": ([[1.0, 1.0], [2.0, -1.0], [1.0, 0.0]], [1.0, 1.0, 2.0], (-1.0, -1.0), (0, False, 2.0),
                              [S(2), S(5), S(1)], [S(3), S(3)]),
    "ruiz :399": ([[1.0, 3.0], [1.0, -2.0], [2.0, 0.0]], [1.0, 1.0, 2.0], (0.0, 0.0), (1, False, None),
                  [S(3), S(2), S(2)], [S(2), S(3)]),
    "ruiz empty row and column :442": ([[2.0, 0.0], [0.0, 0.0]], [1.0, 1.0], (-1.0, -1.0), (1, False, None),
                                       [S(2), 1.0], [S(2), 1.0]),
}


@pytest.mark.parametrize("shards", [1, 2])
@pytest.mark.parametrize("name", sorted(REFERENCE_CASES))
def test_device_rescale_reference_values(gpu_required, name, shards):
    """The values the reference's tests hold for the cumulative rescaling vectors, and
    b / E, c / D, bounds * D for the scaled vectors (preprocess.jl scale_problem)."""
    A, b, lb, (ruiz, l2, alpha), con, var = REFERENCE_CASES[name]
    p = _ref_lp(A, b, lb)
    eng = HipPdhgEngine.from_problem(p, device_ids=[0] * shards) if shards > 1 else HipPdhgEngine.from_problem(p)
    E, D = eng.rescale(ruiz, l2, alpha)
    np.testing.assert_allclose(E, con, rtol=1e-14)
    np.testing.assert_allclose(D, var, rtol=1e-14)
    c, rhs, lo, hi = eng.get_problem_vectors()
    np.testing.assert_allclose(c, np.array([1.0, 2.0]) / var, rtol=1e-14)
    np.testing.assert_allclose(rhs, np.array(b) / con, rtol=1e-14)
    np.testing.assert_allclose(lo, np.array(lb) * var, rtol=1e-14)
    np.testing.assert_allclose(hi, np.array([1.0, 2.0]) * var, rtol=1e-14)
    want = np.array(A) / np.outer(con, var)
    for j in range(2):
        np.testing.assert_allclose(eng.spmv(np.eye(2)[j]), want[:, j], rtol=1e-14, atol=0)
