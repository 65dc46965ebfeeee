"""MPS/QPS input and standard-form transform, mirroring
src/quadratic_programming_io.jl.

The reference delegates parsing to the third-party Julia package QPSReader
0.2.1 (Manifest.toml:117-121), which is not in the reference tree; the free-
and fixed-format MPS grammar is restated here from the published format
(sections NAME, OBJSENSE, ROWS, COLUMNS incl. MARKER lines, RHS, RANGES,
BOUNDS, QUADOBJ/QMATRIX, ENDATA).  Parity for parsing is pinned by the
reference's own fixtures, test/test_qp_io.jl:15-49 -> tests/test_qp_io.py.
"""
import gzip
import math
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

from .quadratic_programming import QuadraticProgrammingProblem, as_csc


@dataclass
class TwoSidedQpProblem:
    """quadratic_programming_io.jl:15-32"""
    variable_lower_bound: np.ndarray
    variable_upper_bound: np.ndarray
    constraint_lower_bound: np.ndarray
    constraint_upper_bound: np.ndarray
    constraint_matrix: sp.csc_matrix
    objective_offset: float
    objective_vector: np.ndarray
    objective_matrix: sp.csc_matrix


def two_sided_rows_to_slacks(qp):
    """quadratic_programming_io.jl:95-131: l <= a'x <= u becomes
    a'x - s = 0, l <= s <= u (in place)."""
    lo, up = qp.constraint_lower_bound, qp.constraint_upper_bound
    two_sided_rows = np.nonzero(np.isfinite(lo) & np.isfinite(up) & (lo != up))[0]
    if len(two_sided_rows) == 0:
        return
    k = len(two_sided_rows)
    m = len(lo)
    slack_matrix = sp.csc_matrix((np.full(k, -1.0), (two_sided_rows, np.arange(k))), shape=(m, k))
    qp.variable_lower_bound = np.concatenate([qp.variable_lower_bound, lo[two_sided_rows]])
    qp.variable_upper_bound = np.concatenate([qp.variable_upper_bound, up[two_sided_rows]])
    qp.objective_vector = np.concatenate([qp.objective_vector, np.zeros(k)])
    qp.constraint_matrix = as_csc(sp.hstack([sp.csc_matrix(qp.constraint_matrix), slack_matrix]))
    qp.constraint_lower_bound = lo.copy()
    qp.constraint_upper_bound = up.copy()
    qp.constraint_lower_bound[two_sided_rows] = 0
    qp.constraint_upper_bound[two_sided_rows] = 0
    n_new = len(qp.variable_lower_bound)
    Q = sp.coo_matrix(qp.objective_matrix)
    qp.objective_matrix = sp.csc_matrix((Q.data, (Q.row, Q.col)), shape=(n_new, n_new))


def transform_to_standard_form(qp):
    """quadratic_programming_io.jl:43-87: equalities first, '<=' rows negated."""
    two_sided_rows_to_slacks(qp)
    lo, up = qp.constraint_lower_bound, qp.constraint_upper_bound
    is_equality_row = lo == up
    is_geq_row = ~is_equality_row & np.isfinite(lo)
    is_leq_row = ~is_equality_row & np.isfinite(up)
    assert not np.any(is_geq_row & is_leq_row)
    num_equalities = int(is_equality_row.sum())
    if num_equalities + int(is_geq_row.sum()) + int(is_leq_row.sum()) != len(lo):
        raise ValueError("Not all constraints have finite bounds on at least one side.")
    A = sp.csr_matrix(qp.constraint_matrix, dtype=np.float64, copy=True)
    sign = np.where(is_leq_row, -1.0, 1.0)
    A = sp.diags(sign) @ A
    new_row_to_old = np.concatenate([np.nonzero(is_equality_row)[0],
                                     np.nonzero(~is_equality_row)[0]])
    A = sp.csr_matrix(A)[new_row_to_old, :]
    right_hand_side = lo.copy()
    right_hand_side[is_leq_row] = -up[is_leq_row]
    right_hand_side = right_hand_side[new_row_to_old]
    return QuadraticProgrammingProblem(
        qp.variable_lower_bound, qp.variable_upper_bound, qp.objective_matrix,
        qp.objective_vector, qp.objective_offset, A, right_hand_side, num_equalities)


# ---- MPS / QPS parsing --------------------------------------------------------

_SECTIONS = {"NAME", "OBJSENSE", "OBJSENSE", "ROWS", "COLUMNS", "RHS", "RANGES",
             "BOUNDS", "QUADOBJ", "QMATRIX", "ENDATA", "OBJSENSE"}


def _fixed_fields(line):
    """Fixed-format MPS columns: 2-3, 5-12, 15-22, 25-36, 40-47, 50-61."""
    spans = [(1, 3), (4, 12), (14, 22), (24, 36), (39, 47), (49, 61)]
    return [line[a:b].strip() for a, b in spans if line[a:b].strip()]


def read_mps(stream, fixed_format=False):
    """Parse an MPS/QPS text stream into a TwoSidedQpProblem."""
    row_kind, row_index = {}, {}
    objective_name = None
    col_index, col_names = {}, []
    a_rows, a_cols, a_vals = [], [], []
    obj = {}
    rhs, ranges = {}, {}
    c0 = 0.0
    lvar, uvar = [], []
    q_rows, q_cols, q_vals, q_full = [], [], [], []   # q_full[k]: triple k came from QMATRIX (both triangles given)
    section = None
    for raw in stream:
        if isinstance(raw, bytes):
            raw = raw.decode("utf-8", "replace")
        line = raw.rstrip("\n").rstrip("\r")
        if not line.strip() or line.lstrip().startswith("*"):
            continue
        if not line[0].isspace():                 # section header
            tok = line.split()
            section = tok[0].upper()
            if section == "ENDATA":
                break
            if section == "OBJSENSE" and len(tok) > 1 and tok[1].upper().startswith("MAX"):
                raise ValueError("OBJSENSE MAX is not supported (the reference asserts objsense == :notset)")
            continue
        f = _fixed_fields(line) if fixed_format else line.split()
        if section == "OBJSENSE":
            if f and f[0].upper().startswith("MAX"):
                raise ValueError("OBJSENSE MAX is not supported")
        elif section == "ROWS":
            kind, name = f[0].upper(), f[1]
            if kind == "N":
                if objective_name is None:
                    objective_name = name
                row_kind[name] = "N"
            else:
                row_kind[name] = kind
                row_index[name] = len(row_index)
        elif section == "COLUMNS":
            if len(f) >= 3 and f[1].upper() == "'MARKER'":
                continue                          # integrality markers are ignored (LP relaxation)
            name = f[0]
            if name not in col_index:
                col_index[name] = len(col_names)
                col_names.append(name)
                lvar.append(0.0)
                uvar.append(math.inf)
            j = col_index[name]
            for r, v in zip(f[1::2], f[2::2]):
                v = float(v)
                if r == objective_name:
                    obj[j] = v
                elif row_kind.get(r) == "N":
                    continue                      # extra free rows are dropped
                else:
                    a_rows.append(row_index[r]); a_cols.append(j); a_vals.append(v)
        elif section == "RHS":
            pairs = f[1:] if len(f) % 2 == 1 else f
            for r, v in zip(pairs[0::2], pairs[1::2]):
                if r == objective_name:
                    c0 = -float(v)                # RHS on the objective row is minus the constant
                elif r in row_index:
                    rhs[row_index[r]] = float(v)
        elif section == "RANGES":
            pairs = f[1:] if len(f) % 2 == 1 else f
            for r, v in zip(pairs[0::2], pairs[1::2]):
                if r in row_index:
                    ranges[row_index[r]] = float(v)
        elif section == "BOUNDS":
            kind = f[0].upper()
            if kind in ("FR", "MI", "PL", "BV"):
                name = f[2] if len(f) >= 3 else f[1]
                val = None
            else:
                name, val = (f[2], float(f[3])) if len(f) >= 4 else (f[1], float(f[2]))
            j = col_index[name]
            if kind == "LO":
                lvar[j] = val
            elif kind == "UP":
                if val < 0 and lvar[j] == 0.0:
                    lvar[j] = -math.inf           # classic MPS rule for a negative upper bound
                uvar[j] = val
            elif kind == "FX":
                lvar[j] = uvar[j] = val
            elif kind == "FR":
                lvar[j], uvar[j] = -math.inf, math.inf
            elif kind == "MI":
                lvar[j] = -math.inf
            elif kind == "PL":
                uvar[j] = math.inf
            elif kind == "BV":
                lvar[j], uvar[j] = 0.0, 1.0
            elif kind == "LI":
                lvar[j] = val
            elif kind == "UI":
                uvar[j] = val
            else:
                raise ValueError(f"unknown bound type {kind}")
        elif section in ("QUADOBJ", "QMATRIX"):
            i, j, v = col_index[f[0]], col_index[f[1]], float(f[2])
            q_rows.append(i); q_cols.append(j); q_vals.append(v)
            q_full.append(section == "QMATRIX")
        elif section in ("NAME", None):
            continue
        else:
            raise ValueError(f"unsupported MPS section {section}")

    ncon, nvar = len(row_index), len(col_names)
    lcon = np.full(ncon, -np.inf)
    ucon = np.full(ncon, np.inf)
    for name, i in row_index.items():
        kind = row_kind[name]
        b = rhs.get(i, 0.0)
        if kind == "E":
            lcon[i] = ucon[i] = b
        elif kind == "G":
            lcon[i] = b
        elif kind == "L":
            ucon[i] = b
        if i in ranges:
            R = ranges[i]
            if kind == "E":
                if R >= 0:
                    ucon[i] = b + abs(R)
                else:
                    lcon[i] = b - abs(R)
            elif kind == "G":
                ucon[i] = b + abs(R)
            elif kind == "L":
                lcon[i] = b - abs(R)
    A = sp.csc_matrix((a_vals, (a_rows, a_cols)), shape=(ncon, nvar), dtype=np.float64)
    c = np.zeros(nvar)
    for j, v in obj.items():
        c[j] = v
    # quadratic_programming_io.jl:166-178: QUADOBJ holds one triangle; mirror it
    qr, qc, qv = [], [], []
    for i, j, v, full in zip(q_rows, q_cols, q_vals, q_full):
        qr.append(i); qc.append(j); qv.append(v)
        if i != j and not full:
            qr.append(j); qc.append(i); qv.append(v)
    Q = sp.csc_matrix((qv, (qr, qc)), shape=(nvar, nvar), dtype=np.float64)
    return TwoSidedQpProblem(np.array(lvar, dtype=np.float64), np.array(uvar, dtype=np.float64),
                             lcon, ucon, A, c0, c, Q)


def qps_reader_to_standard_form(filename, fixed_format=False):
    """quadratic_programming_io.jl:147-197"""
    opener = gzip.open if filename.endswith(".gz") else open
    with opener(filename, "rt") as fh:
        two_sided = read_mps(fh, fixed_format=fixed_format)
    return transform_to_standard_form(two_sided)
