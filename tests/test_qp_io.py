"""test/test_qp_io.jl restated: MPS -> standard form on the reference's two
fixtures (plain and gz), two-sided rows -> slacks; plus BASELINE configs[0]:
solve_qp on trivial_lp_model.mps (CPU plumbing path)."""
import gzip
import json
import os

import numpy as np
import pytest

from firstorderlp_jl_amd.quadratic_programming_io import (
    TwoSidedQpProblem, qps_reader_to_standard_form, two_sided_rows_to_slacks)

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
INF = np.inf


def _check(qp, lb, ub, Q, c, c0, A, b, ne):
    assert np.array_equal(qp.variable_lower_bound, lb)
    assert np.array_equal(qp.variable_upper_bound, ub)
    assert np.array_equal(qp.objective_matrix.toarray(), np.array(Q, dtype=float))
    assert np.array_equal(qp.objective_vector, c)
    assert qp.objective_constant == c0
    assert np.array_equal(qp.constraint_matrix.toarray(), np.array(A, dtype=float))
    assert np.array_equal(qp.right_hand_side, b)
    assert qp.num_equalities == ne


def test_read_mps_lp():      # test_qp_io.jl:15-24, 37-42
    qp = qps_reader_to_standard_form(os.path.join(DATA, "trivial_lp_model.mps"))
    _check(qp, [0.0, 1.0], [1.0, 2.0], np.zeros((2, 2)), [2.0, -1.0], 0.0,
           [[-1.0, -1.0]], [-3.0], 0)


def test_read_mps_qp_and_gz(tmp_path):   # test_qp_io.jl:26-35, 44-63
    path = os.path.join(DATA, "trivial_qp_model.mps")
    for p in (path, None):
        if p is None:
            p = str(tmp_path / "trivial_qp_model.mps.gz")
            with open(path, "rb") as src, gzip.open(p, "wb") as dst:
                dst.write(src.read())
        qp = qps_reader_to_standard_form(p)
        _check(qp, [0.0, 1.0], [1.0, 2.0], [[2.0, 2.0], [2.0, 4.0]], [2.0, -1.0], 0.0,
               [[-1.0, -1.0]], [-3.0], 0)


def test_two_sided_rows_to_slacks():     # test_qp_io.jl:65-94
    import scipy.sparse as sp
    qp = TwoSidedQpProblem(np.array([-INF, -INF]), np.array([INF, INF]),
                           np.array([-3.0, -2.0]), np.array([1.0, INF]),
                           sp.csc_matrix(np.ones((2, 2))), 2.0, np.array([0.0, 1.0]),
                           sp.csc_matrix(np.diag([1.0, 3.0])))
    two_sided_rows_to_slacks(qp)
    assert np.array_equal(qp.variable_lower_bound, [-INF, -INF, -3.0])
    assert np.array_equal(qp.variable_upper_bound, [INF, INF, 1.0])
    assert np.array_equal(qp.constraint_lower_bound, [0.0, -2.0])
    assert np.array_equal(qp.constraint_upper_bound, [0.0, INF])
    assert np.array_equal(qp.constraint_matrix.toarray(), [[1.0, 1.0, -1.0], [1.0, 1.0, 0.0]])
    assert qp.objective_offset == 2.0
    assert np.array_equal(qp.objective_vector, [0.0, 1.0, 0.0])
    assert np.array_equal(qp.objective_matrix.toarray(), np.diag([1.0, 3.0, 0.0]))


def test_ranges_bounds_and_markers(tmp_path):
    mps = """NAME t
ROWS
 N  COST
 N  FREEROW
 E  e1
 G  g1
 L  l1
COLUMNS
    MARKER                 'MARKER'                 'INTORG'
    x  COST 1.0  e1 1.0
    x  g1 2.0  FREEROW 9
    MARKER                 'MARKER'                 'INTEND'
    y  l1 1.0  e1 -1.0
    z  COST -2  g1 1
RHS
    rhs  e1 4  g1 1
    rhs  l1 5  COST -7
RANGES
    rng  e1 -2  g1 3
    rng  l1 4
BOUNDS
 UP bnd x -1
 MI bnd y
 FX bnd z 2.5
ENDATA
"""
    p = tmp_path / "t.mps"
    p.write_text(mps)
    from firstorderlp_jl_amd.quadratic_programming_io import read_mps
    with open(p) as fh:
        q = read_mps(fh)
    assert np.array_equal(q.variable_lower_bound, [-INF, -INF, 2.5])
    assert np.array_equal(q.variable_upper_bound, [-1.0, INF, 2.5])
    assert np.array_equal(q.constraint_lower_bound, [2.0, 1.0, 1.0])   # e1: [4-2,4]; g1: [1,4]; l1: [5-4,5]
    assert np.array_equal(q.constraint_upper_bound, [4.0, 4.0, 5.0])
    assert q.objective_offset == 7.0
    assert np.array_equal(q.objective_vector, [1.0, 0.0, -2.0])
    assert np.array_equal(q.constraint_matrix.toarray(), [[1, -1, 0], [2, 0, 1], [0, 1, 0]])


def test_qmatrix_before_another_section_is_not_mirrored():
    """QMATRIX lists both triangles; a section following it (BOUNDS here) must
    not turn its entries into QUADOBJ-style triangles that get mirrored again."""
    import io
    from firstorderlp_jl_amd.quadratic_programming_io import read_mps
    text = """NAME q
ROWS
 N obj
 G c1
COLUMNS
 x obj 1 c1 1
 y obj 1 c1 1
RHS
 rhs c1 1
QMATRIX
 x x 2
 x y 0.5
 y x 0.5
 y y 3
BOUNDS
 UP bnd x 4
ENDATA
"""
    q = read_mps(io.StringIO(text))
    assert np.array_equal(q.objective_matrix.toarray(), [[2.0, 0.5], [0.5, 3.0]])
    assert q.variable_upper_bound[0] == 4.0
    q2 = read_mps(io.StringIO(text.replace("QMATRIX", "QUADOBJ").replace(" y x 0.5\n", "")))
    assert np.array_equal(q2.objective_matrix.toarray(), [[2.0, 0.5], [0.5, 3.0]])


def test_solve_qp_cli_trivial_lp_cpu_plumbing(tmp_path):
    """BASELINE configs[0]: test/trivial_lp_model.mps, --method pdhg, CPU path.
    Optimum: x = [0, 2], objective -2 (CI.yml:40-45 only requires exit 0)."""
    from scripts import solve_qp
    from tests.oracle_engine import OracleEngine
    out = tmp_path / "out"
    argv = ["--instance_path", os.path.join(DATA, "trivial_lp_model.mps"), "--output_dir", str(out),
            "--method", "pdhg", "--verbosity", "0"]
    output, primal, dual = solve_qp.main(argv, engine_factory=OracleEngine.from_problem)
    assert output.termination_string == "OPTIMAL"
    np.testing.assert_allclose(primal, [0.0, 2.0], atol=1e-5)
    np.testing.assert_allclose(dual, [0.0], atol=1e-5)
    for suffix in ("_summary.json", "_full_log.json.gz", "_primal.txt", "_dual.txt"):
        assert (out / ("trivial_lp_model" + suffix)).exists()
    summary = json.loads((out / "trivial_lp_model_summary.json").read_text())
    assert summary["termination_reason"] == "TERMINATION_REASON_OPTIMAL"
    assert summary["solution_type"] == "POINT_TYPE_AVERAGE_ITERATE"
    assert abs(summary["solution_stats"]["convergence_information"][0]["primal_objective"] + 2.0) < 1e-4
    assert summary["iteration_stats"] == []
    with gzip.open(out / "trivial_lp_model_full_log.json.gz", "rt") as fh:
        assert len(json.load(fh)["iteration_stats"]) >= 1


@pytest.mark.gpu
def test_solve_qp_cli_trivial_models_gpu(gpu_required, tmp_path):
    from scripts import solve_qp
    for name, x_opt in (("trivial_lp_model", [0.0, 2.0]), ("trivial_qp_model", None)):
        out = tmp_path / name
        output, primal, dual = solve_qp.main(
            ["--instance_path", os.path.join(DATA, name + ".mps"), "--output_dir", str(out),
             "--method", "pdhg", "--verbosity", "0"])
        assert output.termination_string == "OPTIMAL"
        if x_opt is not None:
            np.testing.assert_allclose(primal, x_opt, atol=1e-5)
