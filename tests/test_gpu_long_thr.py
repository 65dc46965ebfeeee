"""PDHG_LONG_THR (dev knob, csrc/common.hpp): rows with more entries than the threshold leave the row blocks / the tiled
sweep for the long-row kernels (2 048-entry chunks + ordered combine).  The threshold is kept with each layout
(CsrDev::long_thr) and every consumer of the long-row tables -- products, column slabs, the rescaling passes -- must
agree on it: a row that two paths both claim would get its epilogue, or its scaling, twice."""
import numpy as np
import pytest
import scipy.sparse as sp

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import pagerank_lp
from firstorderlp_jl_amd.preprocess import rescale_problem
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_steps
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _products(eng, p, seed=0, tol=1e-13):
    rng = np.random.default_rng(seed)
    x, y = rng.standard_normal(eng.n), rng.standard_normal(eng.m)
    A = sp.csr_matrix(p.constraint_matrix)
    absA = abs(A)
    ax, aty = eng.spmv(x), eng.spmv_t(y)
    # rows beyond the threshold are summed chunk-wise by their workgroup: 1e-13 of the sum of magnitudes
    np.testing.assert_allclose(ax, A @ x, rtol=0, atol=tol * float((absA @ np.abs(x)).max()))
    np.testing.assert_allclose(aty, A.T @ y, rtol=0, atol=tol * float((absA.T @ np.abs(y)).max()))
    return ax, aty


@pytest.mark.parametrize("spmv", ["stream", "tiled"])
@pytest.mark.parametrize("thr", ["16", "64"])
def test_lowered_threshold_products_steps_and_rescaling(gpu_required, monkeypatch, spmv, thr):
    p = pagerank_lp(30000, seed=4)                      # hub rows and hub columns of several hundred entries
    monkeypatch.setenv("PDHG_SPMV", spmv)
    base = HipPdhgEngine.from_problem(p)
    monkeypatch.setenv("PDHG_LONG_THR", thr)
    eng = HipPdhgEngine.from_problem(p)
    monkeypatch.delenv("PDHG_LONG_THR")
    i0, i1 = base.layout_info(), eng.layout_info()
    assert i1["A_long_rows"] > i0["A_long_rows"] and i1["At_long_rows"] > i0["At_long_rows"]
    _products(eng, p)
    # a trajectory: same accept / reject decisions, iterates to rounding (the hub rows' sums associate differently)
    step, pw = H.initial_step_and_weight(p)
    outs = []
    for e in (base, eng):
        st = PdhgSolverState(e, step_size=step, primal_weight=pw)
        assert take_steps(AdaptiveStepsizeParams(0.3, 0.6), st, 40) == 40
        outs.append((st.total_number_iterations, st.step_size) + tuple(e.get_current()) + tuple(e.get_average()))
    assert outs[0][0] == outs[1][0]
    for a, b in zip(outs[0][1:], outs[1][1:]):
        np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-12)
    # Ruiz rescaling is exact per entry (max-based factors, two multiplications in a fixed order): every resident copy
    # of the matrix must come out bit-identical to the host's, whichever path owns a row -- and scaled exactly once
    host = rescale_problem(10, False, None, 0, p)
    E, D = eng.rescale(10, False, None)
    assert np.array_equal(E, host.constraint_rescaling) and np.array_equal(D, host.variable_rescaling)
    assert eng.matrix_max_abs() == float(np.abs(host.scaled_qp.constraint_matrix.data).max())
    _products(eng, host.scaled_qp, seed=1)
    # the L2 / Pock-Chambolle statistics take the row sums from both kinds of kernel
    eng2 = HipPdhgEngine.from_problem(p)               # default threshold again
    monkeypatch.setenv("PDHG_LONG_THR", thr)
    eng3 = HipPdhgEngine.from_problem(p)
    for e in (eng2, eng3):
        e.rescale(0, True, 1.0)
    a2, t2 = _products(eng2, rescale_problem(0, True, 1.0, 0, p).scaled_qp, seed=2, tol=1e-11)
    a3, t3 = _products(eng3, rescale_problem(0, True, 1.0, 0, p).scaled_qp, seed=2, tol=1e-11)
    np.testing.assert_allclose(a3, a2, rtol=1e-11, atol=1e-11 * np.abs(a2).max())
    np.testing.assert_allclose(t3, t2, rtol=1e-11, atol=1e-11 * np.abs(t2).max())
    for e in (base, eng, eng2, eng3):
        e.close()
