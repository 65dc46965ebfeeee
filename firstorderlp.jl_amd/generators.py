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


def barabasi_albert_edges(num_nodes, degree, seed=0, batch=4096):
    """Preferential-attachment graph in the spirit of
    LightGraphs.barabasi_albert(n, k) (generate_pagerank_lp.jl:121-125): each
    new node attaches to `degree` distinct existing nodes with probability
    proportional to their degree.  Batched (targets are drawn from the endpoint
    list as of the start of the batch) so it is vectorised; not the same RNG
    stream as LightGraphs.  Returns an (E, 2) array of undirected edges."""
    rng = np.random.default_rng(seed)
    k = int(degree)
    n = int(num_nodes)
    assert n > k >= 1
    # seed graph: a path over the first k+1 nodes so every node has degree >= 1
    src = [np.arange(0, k, dtype=np.int64)]
    dst = [np.arange(1, k + 1, dtype=np.int64)]
    endpoints = np.empty(2 * (k + k * (n - k - 1)) + 16, dtype=np.int64)
    ne = 0
    endpoints[ne:ne + k] = src[0]; ne += k
    endpoints[ne:ne + k] = dst[0]; ne += k
    t = k + 1
    while t < n:
        b = min(batch, n - t, max(1, t // 16))   # small batches early: keeps the attachment near-sequential
        idx = rng.integers(0, ne, size=(b, k))
        tgt = np.sort(endpoints[idx], axis=1)
        keep = np.ones((b, k), dtype=bool)
        keep[:, 1:] = tgt[:, 1:] != tgt[:, :-1]          # distinct targets per node
        new = np.repeat(np.arange(t, t + b, dtype=np.int64), k).reshape(b, k)
        s, d = new[keep], tgt[keep]
        src.append(s); dst.append(d)
        endpoints[ne:ne + len(s)] = s; ne += len(s)
        endpoints[ne:ne + len(d)] = d; ne += len(d)
        t += b
    return np.stack([np.concatenate(src), np.concatenate(dst)], axis=1)


def pagerank_lp(num_nodes, approx_num_edges=None, damping_factor=0.99, seed=0):
    """The PageRank LP of benchmarking/generate_pagerank_lp.jl:48-73 in the
    standard form qps_reader_to_standard_form produces (equalities first,
    '<=' rows flipped to '>='):
        sqrt(n) * sum_i x_i                    == sqrt(n)
        x_i - d * sum_{j in N(i)} x_j / deg_j  >= (1 - d) / n      for every i
        x >= 0, objective 0.
    BASELINE configs[2]: num_nodes = 1e6, approx_num_edges = 4e6."""
    n = int(num_nodes)
    if approx_num_edges is None:
        approx_num_edges = 4 * n
    degree = int(round(approx_num_edges / n))
    e = barabasi_albert_edges(n, degree, seed)
    deg = np.bincount(e.reshape(-1), minlength=n).astype(np.float64)
    i = np.concatenate([e[:, 0], e[:, 1]])
    j = np.concatenate([e[:, 1], e[:, 0]])
    S = sp.csr_matrix((-damping_factor / deg[j], (i, j)), shape=(n, n))
    rows = S + sp.identity(n, format="csr")
    top = sp.csr_matrix(np.full((1, n), np.sqrt(n)))
    A = sp.vstack([top, rows], format="csc")
    b = np.concatenate([[np.sqrt(n)], np.full(n, (1.0 - damping_factor) / n)])
    return linear_programming_problem(np.zeros(n), np.full(n, np.inf), np.zeros(n), 0.0,
                                      A, b, 1)


def load_libsvm_file(file_name):
    """benchmarking/generate_l1_svm_lp.jl:106-139: (feature_matrix CSC, labels);
    label 1.0 stays 1.0, anything else becomes -1.0."""
    labels, rows, cols, vals = [], [], [], []
    found_label_one = False
    with open(file_name) as fh:
        for r, line in enumerate(fh):
            parts = line.split()
            if not parts:
                continue
            label = float(parts[0])
            if label == 1.0:
                found_label_one = True
            else:
                label = -1.0
            labels.append(label)
            for tok in parts[1:]:
                c, v = tok.split(":")
                rows.append(len(labels) - 1); cols.append(int(c) - 1); vals.append(float(v))
    assert found_label_one
    X = sp.csc_matrix((vals, (rows, cols)), shape=(len(labels), max(cols) + 1))
    return X, np.array(labels)


def preprocess_training_data(X):
    """generate_l1_svm_lp.jl:141-167: drop empty columns, prepend the dense
    intercept column, scale every column to unit L2 norm."""
    X = sp.csc_matrix(X, dtype=np.float64)
    X = X[:, np.diff(X.indptr) > 0]
    X = sp.hstack([sp.csc_matrix(np.ones((X.shape[0], 1))), X], format="csc")
    norms = np.sqrt(np.asarray(X.multiply(X).sum(axis=0)).reshape(-1))
    norms[norms == 0.0] = 1.0
    return sp.csc_matrix(X @ sp.diags(1.0 / norms))


def l1_svm_lp(feature_matrix, labels, regularizer_weight=1.0):
    """populate_libsvm_model (generate_l1_svm_lp.jl:54-72) in standard form.
    Variables [beta(d) free; w(n) >= 0; z(d) free]; rows, all '>=':
        z - beta >= 0 ; z + beta >= 0 ; w + diag(y) X beta >= 1
    objective sum(w) + regularizer_weight * sum(z)."""
    X = sp.csr_matrix(feature_matrix)
    n, d = X.shape
    y = np.asarray(labels, dtype=np.float64)
    Id, In = sp.identity(d, format="csr"), sp.identity(n, format="csr")
    Zdn, Znd = sp.csr_matrix((d, n)), sp.csr_matrix((n, d))
    A = sp.bmat([[-Id, Zdn, Id], [Id, Zdn, Id], [sp.diags(y) @ X, In, Znd]], format="csc")
    b = np.concatenate([np.zeros(2 * d), np.ones(n)])
    c = np.concatenate([np.zeros(d), np.ones(n), np.full(d, float(regularizer_weight))])
    lb = np.concatenate([np.full(d, -np.inf), np.zeros(n), np.full(d, -np.inf)])
    ub = np.full(2 * d + n, np.inf)
    return linear_programming_problem(lb, ub, c, 0.0, A, b, 0)


def synthetic_rcv1_like(num_samples=20242, num_features=47236, nnz_per_row=74, seed=0):
    """A LIBSVM-rcv1.binary-shaped training set (the file itself is not
    available offline): Zipf-distributed feature frequencies, positive
    tf-idf-like values, rows scaled to unit L2 norm, labels from a planted
    sparse separator with 5% label noise."""
    rng = np.random.default_rng(seed)
    N, D, k = int(num_samples), int(num_features), int(nnz_per_row)
    weights = 1.0 / np.arange(1, D + 1) ** 0.9
    weights /= weights.sum()
    counts = np.maximum(1, rng.poisson(k, size=N))
    rows = np.repeat(np.arange(N), counts)
    cols = rng.choice(D, size=int(counts.sum()), p=weights)
    vals = rng.gamma(2.0, 1.0, size=len(cols))
    X = sp.csr_matrix((vals, (rows, cols)), shape=(N, D))
    X.sum_duplicates()
    rn = np.sqrt(np.asarray(X.multiply(X).sum(axis=1)).reshape(-1))
    X = sp.diags(1.0 / np.maximum(rn, 1e-300)) @ X
    w = np.zeros(D)
    sup = rng.choice(D, size=min(200, D), replace=False, p=weights)
    w[sup] = rng.standard_normal(len(sup))
    score = X @ w
    labels = np.where(score >= np.median(score), 1.0, -1.0)
    flip = rng.random(N) < 0.05
    labels[flip] *= -1
    return sp.csc_matrix(X), labels


def l1_svm_rcv1_like_lp(num_samples=20242, num_features=47236, nnz_per_row=74,
                        regularizer_weight=1.0, seed=0):
    """BASELINE configs[3] on the synthetic rcv1-shaped data."""
    X, labels = synthetic_rcv1_like(num_samples, num_features, nnz_per_row, seed)
    return l1_svm_lp(preprocess_training_data(X), labels, regularizer_weight)
