"""Seeded synthetic LP generators for the benchmark configs.

The reference's own generators (benchmarking/generate_pagerank_lp.jl,
generate_l1_svm_lp.jl) build models through JuMP/LightGraphs RNG streams that
cannot be replayed without Julia; these restate the *models*, not the streams.
"""
import numpy as np
import scipy.sparse as sp

from .quadratic_programming import linear_programming_problem


def random_lp(m, n, nnz_per_row=10, seed=12345):
    """BASELINE config 4: each row draws ``nnz_per_row`` distinct columns
    uniformly, values N(0,1); feasible and bounded by construction
    (x0 primal feasible, y0 dual feasible).  First m//2 rows are equalities."""
    rng = np.random.default_rng(seed)
    k = int(nnz_per_row)
    cols = rng.integers(0, n, size=(m, k), dtype=np.int32)
    cols.sort(axis=1)
    dup_rows = np.nonzero((cols[:, 1:] == cols[:, :-1]).any(axis=1))[0]
    for r in dup_rows:  # rare: redraw without replacement
        cols[r] = np.sort(rng.choice(n, size=k, replace=False)).astype(np.int32)
    vals = rng.standard_normal((m, k))
    indptr = np.arange(0, m * k + 1, k, dtype=np.int64)
    A = sp.csr_matrix((vals.reshape(-1), cols.reshape(-1), indptr), shape=(m, n))
    num_eq = m // 2
    x0 = rng.random(n)
    at_lower = rng.random(n) < 0.3
    x0[at_lower] = 0.0
    y0 = rng.standard_normal(m)
    y0[num_eq:] = np.abs(y0[num_eq:])
    b = A @ x0
    slack = rng.random(m - num_eq)
    tight = rng.random(m - num_eq) < 0.5
    slack[tight] = 0.0
    y0[num_eq:][~tight] = 0.0
    b[num_eq:] -= slack
    lb = np.zeros(n)
    ub = np.full(n, np.inf)
    ub[rng.random(n) < 0.2] = 10.0
    r = np.zeros(n)
    r[at_lower] = rng.random(int(at_lower.sum()))
    c = A.T @ y0 + r
    return linear_programming_problem(lb, ub, c, 0.0, A.tocsc(), b, num_eq)
