"""test/test_qp_processing.jl restated (the exact-value rescaling / presolve
cases): pins firstorderlp.jl_amd/preprocess.py.  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

from firstorderlp_jl_amd import QuadraticProgrammingProblem, linear_programming_problem
from firstorderlp_jl_amd import preprocess as P

S = np.sqrt


def _lp(lb, ub, c, A, b, ne):
    return linear_programming_problem(lb, ub, c, 0.0, A, b, ne)


def _same(p, q, tol=1e-12):
    for f in ("variable_lower_bound", "variable_upper_bound", "objective_vector", "right_hand_side"):
        np.testing.assert_allclose(getattr(p, f), getattr(q, f), rtol=tol, atol=tol)
    np.testing.assert_allclose(p.constraint_matrix.toarray(), q.constraint_matrix.toarray(), rtol=tol, atol=tol)
    np.testing.assert_allclose(p.objective_matrix.toarray(), q.objective_matrix.toarray(), rtol=tol, atol=tol)
    assert p.num_equalities == q.num_equalities and abs(p.objective_constant - q.objective_constant) <= tol


def test_l2_norm():                                  # :16-20
    M = sp.csc_matrix(np.array([[3.0, 0.0, -4.0], [4.0, 3.0, 0.0]]))
    np.testing.assert_allclose(P.l2_norm(M, 1), [5.0, 3.0, 4.0], atol=1e-10)
    np.testing.assert_allclose(P.l2_norm(M, 2), [5.0, 5.0], atol=1e-10)


def test_l2_norm_rescaling_lp():                     # :234-267
    p = _lp([0.0, 0.0], [1.0, 2.0], [1.0, 2.0], [[1.0, 1.0], [1.0, -1.0], [1.0, 0.0]], [1.0, 1.0, 2.0], 1)
    P.l2_norm_rescaling(p)
    q = _lp([0.0, 0.0], [3 ** 0.25, 2.0 * 2 ** 0.25], [1.0 / 3 ** 0.25, 2.0 / 2 ** 0.25],
            [[6 ** -0.25, 4 ** -0.25], [6 ** -0.25, -(4 ** -0.25)], [3 ** -0.25, 0.0]],
            [2 ** -0.25, 2 ** -0.25, 2.0], 1)
    _same(p, q)


@pytest.mark.parametrize("alpha,con,var", [(0.0, [S(2), S(2), S(2)], [S(6), S(2)]),
                                           (1.0, [S(2), S(3), S(1)], [S(4), S(2)]),
                                           (2.0, [S(2), S(5), S(1)], [S(3), S(3)])])
def test_pock_chambolle(alpha, con, var):            # :339-397
    p = _lp([-1.0, -1.0], [1.0, 2.0], [1.0, 2.0], [[1.0, 1.0], [2.0, -1.0], [1.0, 0.0]], [1.0, 1.0, 2.0], 1)
    cr, vr = P.pock_chambolle_rescaling(p, alpha)
    np.testing.assert_allclose(cr, con, rtol=1e-14)
    np.testing.assert_allclose(vr, var, rtol=1e-14)


def test_ruiz_rescaling_lp_one_iteration():          # :399-440
    p = _lp([0.0, 0.0], [1.0, 2.0], [1.0, 2.0], [[1.0, 3.0], [1.0, -2.0], [2.0, 0.0]], [1.0, 1.0, 2.0], 1)
    original = p.copy()
    cr, vr = P.ruiz_rescaling(p, 1)
    q = _lp([0.0, 0.0], [S(2), 2.0 * S(3)], [1.0 / S(2), 2.0 / S(3)],
            [[1 / S(6), 1.0], [0.5, -S(2) / S(3)], [1.0, 0.0]], [1 / S(3), 1 / S(2), S(2)], 1)
    _same(p, q)
    np.testing.assert_allclose(vr, [2 ** 0.5, 3 ** 0.5], rtol=1e-14)
    np.testing.assert_allclose(cr, [3 ** 0.5, 2 ** 0.5, 2 ** 0.5], rtol=1e-14)
    P.unscale_problem(p, cr, vr)
    _same(p, original)


def test_ruiz_with_empty_row_and_column():           # :442-481
    p = _lp([-1.0, -1.0], [1.0, 2.0], [1.0, 2.0], [[2.0, 0.0], [0.0, 0.0]], [1.0, 1.0], 1)
    original = p.copy()
    cr, vr = P.ruiz_rescaling(p, 1)
    q = _lp([-S(2), -1.0], [S(2), 2.0], [1 / S(2), 2.0], [[1.0, 0.0], [0.0, 0.0]], [1 / S(2), 1.0], 1)
    _same(p, q)
    np.testing.assert_allclose(vr, [S(2), 1.0], rtol=1e-14)
    np.testing.assert_allclose(cr, [S(2), 1.0], rtol=1e-14)
    P.unscale_problem(p, cr, vr)
    _same(p, original)


def test_ruiz_converges_and_rescale_problem_roundtrip():   # :483-547
    A = [[1.0, 3.0], [1.0, -2.0], [2.0, 0.0]]
    p = _lp([0.0, 0.0], [1.0, 2.0], [1.0, 2.0], A, [1.0, 1.0, 3.0], 1)
    original = p.copy()
    cr, vr = P.ruiz_rescaling(p, 30)
    M = np.abs(p.constraint_matrix.toarray())
    np.testing.assert_allclose(np.sqrt(M.max(axis=0)), 1.0, rtol=1e-7)
    np.testing.assert_allclose(np.sqrt(M.max(axis=1)), 1.0, rtol=1e-7)
    P.unscale_problem(p, cr, vr)
    _same(p, original)
    sp_ = P.rescale_problem(10, True, None, 0, original)
    _same(sp_.original_qp, original, tol=0)             # original untouched
    P.unscale_problem(sp_.scaled_qp, sp_.constraint_rescaling, sp_.variable_rescaling)
    _same(sp_.scaled_qp, original)


def test_ruiz_qp():                                  # :549-592 (structure: Q scaled by D^-1 Q D^-1)
    p = QuadraticProgrammingProblem([-np.inf, -2.0], [1.0, 2.0], [[4.0, 2.0], [2.0, 1.0]], [1.0, 2.0], 0.0,
                                    [[1.0, 3.0], [1.0, -2.0], [2.0, 0.0]], [1.0, 1.0, 2.0], 1)
    original = p.copy()
    cr, vr = P.ruiz_rescaling(p, 1)
    np.testing.assert_allclose(vr, [2.0, S(3)], rtol=1e-14)     # sqrt(max(col max |A|, col max |Q|))
    np.testing.assert_allclose(p.objective_matrix.toarray(),
                               np.array([[4.0, 2.0], [2.0, 1.0]]) / np.outer(vr, vr), rtol=1e-14)
    P.unscale_problem(p, cr, vr)
    _same(p, original)


def test_remove_empty_rows_and_columns_and_presolve():      # :22-232
    p = _lp([0.0, 0.0, -1.0], [1.0, 2.0, 3.0], [1.0, 2.0, -5.0],
            [[1.0, 3.0, 0.0], [0.0, 0.0, 0.0], [2.0, 0.0, 0.0]], [1.0, -1.0, 3.0], 1)
    info = P.presolve(p, verbosity=0)
    assert list(info.empty_rows) == [1] and list(info.empty_columns) == [2]
    assert p.constraint_matrix.shape == (2, 2) and p.num_equalities == 1
    assert p.objective_constant == -15.0            # removed variable at its upper bound (coef < 0)
    x, y = P.undo_presolve(info, np.array([0.5, 0.25]), np.array([1.0, 2.0]))
    np.testing.assert_allclose(x, [0.5, 0.25, 0.0])  # eliminated entries are 0, then projected
    np.testing.assert_allclose(y, [1.0, 0.0, 2.0])
    bad = _lp([0.0], [1.0], [1.0], [[0.0], [1.0]], [1.0, 1.0], 1)   # empty equality row with rhs != 0
    with pytest.raises(ValueError):
        P.remove_empty_rows(bad)
    bad2 = _lp([0.0], [1.0], [1.0], [[1.0], [0.0]], [1.0, 1.0], 1)  # empty inequality row with rhs > 0
    with pytest.raises(ValueError):
        P.remove_empty_rows(bad2)


# ---- the remaining testsets of test/test_qp_processing.jl, one function per testset ----

def test_remove_empty_rows_inequality():             # :22-52
    p = _lp([0.0, 0.0], [1.0, 2.0], [1.0, 2.0], [[2.0, 0.0], [1.0, 0.0], [0.0, 0.0]], [1.0, 1.0, -1.0], 1)
    P.remove_empty_rows(p)
    _same(p, _lp([0.0, 0.0], [1.0, 2.0], [1.0, 2.0], [[2.0, 0.0], [1.0, 0.0]], [1.0, 1.0], 1), tol=0)


def test_remove_empty_rows_equality():               # :54-84
    p = _lp([0.0, 0.0], [1.0, 2.0], [1.0, 2.0], [[0.0, 0.0], [1.0, 0.0], [1.0, 0.0]], [0.0, 1.0, 0.0], 1)
    P.remove_empty_rows(p)
    _same(p, _lp([0.0, 0.0], [1.0, 2.0], [1.0, 2.0], [[1.0, 0.0], [1.0, 0.0]], [1.0, 0.0], 0), tol=0)


def test_remove_empty_rows_errors_on_positive_inequality_rhs():   # :86-102
    p = _lp([0.0, 0.0], [1.0, 2.0], [1.0, 2.0], [[1.0, 0.0], [1.0, 0.0], [0.0, 0.0]], [1.0, 1.0, 1.0], 1)
    with pytest.raises(ValueError):
        P.remove_empty_rows(p)


def test_remove_empty_rows_errors_on_nonzero_equality_rhs():      # :104-120
    p = _lp([0.0, 0.0], [1.0, 2.0], [1.0, 2.0], [[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]], [1.0, 1.0, 1.0], 1)
    with pytest.raises(ValueError):
        P.remove_empty_rows(p)


@pytest.mark.parametrize("c0,constant", [(3.0, -3.0), (-3.0, -6.0)])   # :122-148 lower bound, :150-176 upper
def test_remove_empty_columns(c0, constant):
    p = _lp([-1.0, -1.0], [2.0, 2.0], [c0, 2.0], [[0.0, 1.0], [0.0, -1.0]], [1.0, 1.0], 0)
    P.remove_empty_columns(p)
    q = linear_programming_problem([-1.0], [2.0], [2.0], constant, [[1.0], [-1.0]], [1.0, 1.0], 0)
    _same(p, q, tol=0)


def test_recover_original_solution():                # :178-188 (1-based [1, 4] there)
    out = P.recover_original_solution(np.array([1.0, 1.0, 1.0, 5.0]), np.array([0, 3]), 5)
    assert list(out) == [0.0, 1.0, 1.0, 0.0, 1.0]


def test_presolve_then_undo():                       # :190-209
    p = _lp([0.0, 0.0, 1.0], [1.0, 2.0, 2.0], [1.0, 2.0, 0.0],
            [[1.0, 1.0, 0.0], [1.0, -1.0, 0.0], [0.0, 0.0, 0.0]], [1.0, 1.0, 0.0], 1)
    info = P.presolve(p, verbosity=0)
    x, y = P.undo_presolve(info, np.array([1.0, 0.0]), np.array([1.0, 1.0]))
    assert list(x) == [1.0, 0.0, 1.0]                # the removed variable lands on its lower bound
    assert list(y) == [1.0, 1.0, 0.0]


def test_presolve_keeps_empty_columns_of_a_qp():     # :211-232
    p = QuadraticProgrammingProblem([0.0, 0.0, 0.0], [1.0, 2.0, 1.0],
                                    [[4.0, 2.0, 0.0], [2.0, 1.0, 0.0], [0.0, 0.0, 1.0]], [1.0, 2.0, 1.0], 0.0,
                                    [[1.0, 1.0, 0.0], [1.0, -1.0, 0.0], [1.0, 0.0, 0.0]], [1.0, 1.0, 2.0], 1)
    P.presolve(p, verbosity=0)
    assert p.constraint_matrix.shape == (3, 3)


def test_l2_norm_rescaling_with_empty_rows():        # :269-302
    p = _lp([0.0, 0.0], [1.0, 2.0], [1.0, 2.0], [[1.0, 1.0], [1.0, -1.0], [0.0, 0.0]], [1.0, 1.0, 0.0], 1)
    P.l2_norm_rescaling(p)
    r = 2 ** 0.25
    q = _lp([0.0, 0.0], [r, 2.0 * r], [1.0 / r, 2.0 / r],
            [[4 ** -0.25, 4 ** -0.25], [4 ** -0.25, -(4 ** -0.25)], [0.0, 0.0]], [1 / r, 1 / r, 0.0], 1)
    _same(p, q)


def test_l2_norm_rescaling_with_empty_columns():     # :304-337
    p = _lp([0.0, 0.0], [1.0, 2.0], [1.0, 2.0], [[1.0, 0.0], [1.0, 0.0], [2.0, 0.0]], [1.0, 1.0, 2.0], 1)
    P.l2_norm_rescaling(p)
    q = _lp([0.0, 0.0], [6 ** 0.25, 2.0], [6 ** -0.25, 2.0],
            [[6 ** -0.25, 0.0], [6 ** -0.25, 0.0], [24 ** -0.25 * 2.0, 0.0]], [1.0, 1.0, 2.0 / S(2)], 1)
    _same(p, q)


def _qp_case(lb0):
    return QuadraticProgrammingProblem([lb0, -2.0], [1.0, 2.0], [[4.0, 2.0], [2.0, 1.0]], [1.0, 2.0], 0.0,
                                       [[1.0, 3.0], [1.0, -2.0], [2.0, 0.0]], [1.0, 1.0, 2.0], 1)


def test_ruiz_qp_exact_values():                     # :549-592
    p = _qp_case(-np.inf)
    cr, vr = P.ruiz_rescaling(p, 1)
    np.testing.assert_allclose(vr, [2.0, S(3)], rtol=1e-14)
    np.testing.assert_allclose(cr, [S(3), S(2), S(2)], rtol=1e-14)
    q = QuadraticProgrammingProblem([-np.inf, -2 * S(3)], [2.0, 2 * S(3)],
                                    [[1.0, 1 / S(3)], [1 / S(3), 1 / 3.0]], [0.5, 2 / S(3)], 0.0,
                                    [[1 / (2 * S(3)), 1.0], [1 / (2 * S(2)), -S(2) / S(3)], [1 / S(2), 0.0]],
                                    [1 / S(3), 1 / S(2), S(2)], 1)
    _same(p, q)


def test_ruiz_qp_converges():                        # :594-633
    p = _qp_case(-1.0)
    original = p.copy()
    cr, vr = P.ruiz_rescaling(p, 30)
    A, Q = np.abs(p.constraint_matrix.toarray()), np.abs(p.objective_matrix.toarray())
    np.testing.assert_allclose(np.sqrt(np.maximum(A.max(axis=0), Q.max(axis=0))), 1.0, rtol=1e-7)
    np.testing.assert_allclose(np.sqrt(A.max(axis=1)), 1.0, rtol=1e-7)
    P.unscale_problem(p, cr, vr)
    _same(p, original)


def test_l2_ruiz_lp_one_iteration():                 # :635-671
    p = _lp([0.0, 0.0], [1.0, 2.0], [1.0, 2.0], [[1.0, 3.0], [1.0, -2.0], [2.0, 0.0]], [1.0, 1.0, 3.0], 1)
    cr, vr = P.ruiz_rescaling(p, 1, 2.0)
    q = _lp([0.0, 0.0], [6 ** 0.25, 2 * 13 ** 0.25], [6 ** -0.25, 2 / 13 ** 0.25],
            [[(6 * 15) ** -0.25, 3 / (13 * 15) ** 0.25], [(7.5 * 6) ** -0.25, -2 / (13 * 7.5) ** 0.25],
             [2 / 36 ** 0.25, 0.0]], [15 ** -0.25, 7.5 ** -0.25, 3 / 6 ** 0.25], 1)
    _same(p, q)
    np.testing.assert_allclose(vr, [6 ** 0.25, 13 ** 0.25], rtol=1e-14)
    np.testing.assert_allclose(cr, [15 ** 0.25, 7.5 ** 0.25, 6 ** 0.25], rtol=1e-14)


def test_l2_ruiz_lp_converges():                     # :672-692
    p = _lp([0.0, 0.0], [1.0, 2.0], [1.0, 2.0], [[1.0, 3.0], [1.0, -2.0], [2.0, 0.0]], [1.0, 1.0, 3.0], 1)
    P.ruiz_rescaling(p, 60, 2.0)
    np.testing.assert_allclose(P.l2_norm(p.constraint_matrix, 1), [1.0, 1.0], atol=1e-5)
    np.testing.assert_allclose(P.l2_norm(p.constraint_matrix, 2), [S(2 / 3)] * 3, atol=1e-5)


def test_l2_ruiz_qp_one_iteration():                 # :694-733
    p = _qp_case(-np.inf)
    cr, vr = P.ruiz_rescaling(p, 1, 2.0)
    q = QuadraticProgrammingProblem(
        [-np.inf, -2 * 18 ** 0.25], [26 ** 0.25, 2 * 18 ** 0.25],
        [[4 / 26 ** 0.5, 2 / (26 * 18) ** 0.25], [2 / (26 * 18) ** 0.25, 1 / 18 ** 0.5]],
        [26 ** -0.25, 2 / 18 ** 0.25], 0.0,
        [[(25 * 26) ** -0.25, 3 / (18 * 25) ** 0.25], [(12.5 * 26) ** -0.25, -2 / (18 * 12.5) ** 0.25],
         [2 / (10 * 26) ** 0.25, 0.0]], [25 ** -0.25, 12.5 ** -0.25, 2 / 10 ** 0.25], 1)
    _same(p, q)
    np.testing.assert_allclose(vr, [26 ** 0.25, 18 ** 0.25], rtol=1e-14)
    np.testing.assert_allclose(cr, [25 ** 0.25, 12.5 ** 0.25, 10 ** 0.25], rtol=1e-14)


def test_l2_ruiz_qp_converges():                     # :735-766
    p = _qp_case(-1.0)
    P.ruiz_rescaling(p, 100, 2.0)
    cols = np.sqrt(np.sqrt(P.l2_norm(p.constraint_matrix, 1) ** 2 + P.l2_norm(p.objective_matrix, 1) ** 2))
    np.testing.assert_allclose(cols, [1.0, 1.0], atol=1e-5)
    np.testing.assert_allclose(P.l2_norm(p.constraint_matrix, 2), [S(2 / 5)] * 3, atol=1e-5)


def test_l2_ruiz_closed_form():                      # :770-800
    p = _lp([0.0, 0.0], [1.0, 2.0], [1.0, 2.0], [[1.0, 1.0], [1.0, -1.0], [1.0, 1.0]], [1.0, 1.0, 3.0], 1)
    P.ruiz_rescaling(p, 10, 2.0)
    r, t = 3 ** 0.25, 1 / S(3)
    _same(p, _lp([0.0, 0.0], [r, 2 * r], [1 / r, 2 / r], [[t, t], [t, -t], [t, t]], [1 / r, 1 / r, 3 / r], 1))


@pytest.mark.parametrize("before,perm,after", [                     # test_sparse_linalg.jl:16-20, 22-35
    ([[1.0, 0.0], [0.0, 1.0]], [1, 0], [[0.0, 1.0], [1.0, 0.0]]),
    ([[1.0, 0.0], [0.0, 1.0], [2.0, 3.0]], [2, 0, 1], [[0.0, 1.0], [2.0, 3.0], [1.0, 0.0]])])
def test_row_permute_in_place(before, perm, after):
    M = sp.csc_matrix(np.array(before))
    data_buffer, index_buffer = M.data, M.indices
    P.row_permute_in_place(M, perm)
    assert M.data is data_buffer and M.indices is index_buffer      # in place
    assert np.array_equal(M.toarray(), np.array(after))
    assert np.all(np.diff(M.indices[M.indptr[1]:M.indptr[2]]) > 0)
