#!/usr/bin/env python3
"""Both fused products (A xbar with the dual step; A'y' with the interaction sums) on a range of matrix shapes, with the
vendor's CSR SpMV (rocSPARSE, best of its four CSR algorithms, tools/vendor_spmv.py) on the same device and the same
CSR arrays beside every product.  Measurement aid: prints one line per shape (profiles/r05_shape_table.txt).

    python tools/shape_table.py [--only name,name] [--no-vendor] [--env "K=V K=V"]
"""
import argparse
import os
os.environ.setdefault("PDHG_DEV", "1")     # development variables on (csrc/common.hpp: dev_env)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import folp_loader  # noqa: E402

folp_loader.load()

SHAPES = [
    ("uniform 1M x 1M, 10 per row", "random", dict(m=1_000_000, n=1_000_000)),
    ("uniform 4M x 4M, 10 per row", "random", dict(m=4_000_000, n=4_000_000)),
    ("uniform 10M x 10M, 10 per row (config S)", "random", {}),
    ("uniform 10M x 10M, 30 per row", "random", dict(k=30)),
    ("tall 10M x 1M, 10 per row", "random", dict(n=1_000_000)),
    ("column-skewed 10M", "colskew", {}),
    ("banded 10M +-50000", "banded", dict(band=50_000)),
    ("banded 10M +-3000000", "banded", dict(band=3_000_000)),
    ("blockdiag 10M", "blockdiag", {}),
    ("clustered 10M", "clustered", {}),
    ("twodensity 10M", "twodensity", {}),
    ("arrowhead 10M", "arrowhead", {}),
    ("pagerank 1M", "pagerank", dict(n=1_000_000)),
    ("l1svm rcv1-shaped SUBSTITUTE", "l1svm", {}),
]
# the structured shapes at other sizes (--only "1M-", "4M-", "200K-": never part of the default table): one residency round,
# short waves, vectors of 1.6 / 8 / 32 MB
for _tag, _n in (("200K", 200_000), ("1M", 1_000_000), ("4M", 4_000_000)):
    SHAPES += [
        (f"{_tag}-uniform", "random", dict(m=_n, n=_n)),
        (f"{_tag}-column-skewed", "colskew", dict(m=_n, n=_n)),
        (f"{_tag}-banded +-{_n // 200}", "banded", dict(m=_n, n=_n, band=_n // 200)),
        (f"{_tag}-banded +-{3 * _n // 10}", "banded", dict(m=_n, n=_n, band=3 * _n // 10)),
        (f"{_tag}-blockdiag", "blockdiag", dict(m=_n, n=_n)),
        (f"{_tag}-clustered", "clustered", dict(m=_n, n=_n)),
        (f"{_tag}-twodensity", "twodensity", dict(m=_n, n=_n)),
        (f"{_tag}-arrowhead", "arrowhead", dict(m=_n, n=_n)),
        (f"{_tag}-wide {3 * _n // 10} x {3 * _n}", "random", dict(m=3 * _n // 10, n=3 * _n)),
        (f"{_tag}-30 per row", "random", dict(m=_n, n=_n, k=30)),
        (f"{_tag}-lognormal rows", "lognormal", dict(m=_n, n=_n)),
    ]
SHAPES += [("10M-linking rows", "manylong", {}), ("1M-linking rows", "manylong", dict(m=1_000_000, n=1_000_000)),
           ("4M-pagerank", "pagerank", dict(n=4_000_000)), ("10M-lognormal rows", "lognormal", {}), ("10M-banded +-50000, lognormal rows", "bandlog", dict(band=50_000)),
           ("1M-banded +-5000, lognormal rows", "bandlog", dict(m=1_000_000, n=1_000_000, band=5_000))]


def make_shape(kind, m=10_000_000, n=10_000_000, k=10, band=0):
    """The structured test matrices of tools/tune_tiled.py (same seeds) as LP problems.
    SHAPE_CACHE_DIR: keep the generated problems there (pickle) -- the counter passes of tools/pmc_stream.sh run the same
    shape many times and a 100M-nonzero matrix takes a minute to generate."""
    import pickle
    cache = os.environ.get("SHAPE_CACHE_DIR")
    path = os.path.join(cache, f"shape_{kind}_{m}_{n}_{k}_{band}.pkl") if cache else None
    if path and os.path.exists(path):
        with open(path, "rb") as fh:
            return pickle.load(fh)
    p = _make_shape(kind, m, n, k, band)
    if path:
        os.makedirs(cache, exist_ok=True)
        with open(path + ".tmp", "wb") as fh:
            pickle.dump(p, fh, protocol=pickle.HIGHEST_PROTOCOL)
        os.replace(path + ".tmp", path)
    return p


def _make_shape(kind, m, n, k, band):
    import scipy.sparse as sp
    from firstorderlp_jl_amd import linear_programming_problem
    from firstorderlp_jl_amd.generators import l1_svm_rcv1_like_lp, pagerank_lp, random_lp
    if kind == "random":
        return random_lp(m, n, k, 12345)
    if kind == "pagerank":
        return pagerank_lp(n, seed=1)
    if kind == "l1svm":
        return l1_svm_rcv1_like_lp(seed=0)
    seed = {"colskew": 5, "banded": 6}.get(kind, 7)
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(m, dtype=np.int64), k)
    if kind == "colskew":               # column popularity ~ 1/sqrt(index)
        cols = np.minimum((rng.random(m * k) ** 2 * n).astype(np.int64), n - 1)
    elif kind == "banded":              # row i: k entries within +-band columns of i*n/m
        centre = (np.arange(m, dtype=np.int64) * n) // m
        cols = np.clip(np.repeat(centre, k) + rng.integers(-band, band + 1, m * k), 0, n - 1)
    elif kind == "blockdiag":           # 100 diagonal blocks, uniform inside a block
        nb = 100
        cols = (rows * nb // m) * (n // nb) + rng.integers(0, n // nb, m * k)
    elif kind == "clustered":           # every row: k entries within +-500 columns of a random centre
        centre = np.repeat(rng.integers(0, n, m), k)
        cols = np.clip(centre + rng.integers(-500, 501, m * k), 0, n - 1)
    elif kind == "twodensity":          # alternating rows of k/3 and 5k/3 entries
        lens = np.where(np.arange(m) % 2 == 0, max(1, k // 3), 5 * k // 3)
        rows = np.repeat(np.arange(m, dtype=np.int64), lens)
        cols = rng.integers(0, n, rows.size)
    elif kind == "lognormal":           # row lengths exp(N(log k - 0.5, 1)) (mean ~k, a tail of rows with hundreds of entries), uniform columns
        lens = np.clip(np.exp(rng.normal(np.log(k) - 0.5, 1.0, m)).astype(np.int64), 1, 4000)
        rows = np.repeat(np.arange(m, dtype=np.int64), lens)
        cols = rng.integers(0, n, rows.size)
    elif kind == "bandlog":             # log-normal row lengths, columns within +-band of the row's own position
        lens = np.clip(np.exp(rng.normal(np.log(k) - 0.5, 1.0, m)).astype(np.int64), 1, 4000)
        rows = np.repeat(np.arange(m, dtype=np.int64), lens)
        cols = np.clip((rows * n) // m + rng.integers(-band, band + 1, rows.size), 0, n - 1)
    elif kind == "manylong":            # k/2 per row uniform + m/5000 linking rows of 2.5 k * 1000 entries each (half the nonzeros in long rows)
        kk = max(1, k // 2)
        rows = np.repeat(np.arange(m, dtype=np.int64), kk)
        cols = rng.integers(0, n, rows.size)
        nl, ll = max(1, m // 5000), 2500 * k
        lr = np.repeat(rng.choice(m, nl, replace=False), ll)
        rows = np.concatenate([rows, lr])
        cols = np.concatenate([cols, rng.integers(0, n, lr.size)])
    elif kind == "arrowhead":           # uniform + 5 dense rows + 5 dense columns
        cols = rng.integers(0, n, m * k)
        dr = np.repeat(np.arange(5, dtype=np.int64) * (m // 5), n // 4)
        dc = np.tile(rng.choice(n, n // 4, replace=False), 5)
        er = rng.choice(m, m // 4, replace=False)
        rows = np.concatenate([rows, dr, np.tile(er, 5)])
        cols = np.concatenate([cols, dc, np.repeat(np.arange(5, dtype=np.int64) * (n // 5) + 1, er.size)])
    else:
        raise SystemExit(f"unknown shape {kind}")
    M = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(m, n))
    M.sum_duplicates()
    return linear_programming_problem(np.zeros(n), np.full(n, 10.0), rng.standard_normal(n), 0.0,
                                      M.tocsc(), rng.standard_normal(m), m // 2)


def product_ms(p, steps=30):
    """HIP-event brackets around the two fused products over `steps` adaptive take_steps (as tools/tune_tiled.py)."""
    import folp_loader
    pkg = folp_loader.load()
    from firstorderlp_jl_amd import _lib
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_step
    A = p.constraint_matrix
    step0 = 1.0 / float(np.abs(A.data).max())
    cn, bn = np.linalg.norm(p.objective_vector), np.linalg.norm(p.right_hand_side)
    pw0 = float(cn / bn) if cn > 0 and bn > 0 else 1.0
    t0 = time.time()
    eng = pkg.HipPdhgEngine.from_problem(p)
    tc = time.time() - t0
    st = PdhgSolverState(eng, step_size=step0, primal_weight=pw0)
    for _ in range(5):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
    eng.profile_enable(True)
    for _ in range(steps):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
    c1, m1 = eng.profile_read(_lib.K_SPMV_DUAL)
    c2, m2 = eng.profile_read(_lib.K_SPMV_ATY)
    out = dict(dual_ms=m1 / c1, aty_ms=m2 / c2, dual_bytes=eng.kernel_algorithmic_bytes(_lib.K_SPMV_DUAL),
               aty_bytes=eng.kernel_algorithmic_bytes(_lib.K_SPMV_ATY), create_s=tc,
               kernels=(eng.kernel_name(_lib.K_SPMV_DUAL), eng.kernel_name(_lib.K_SPMV_ATY)))
    info = eng.layout_info()
    out["waves"], out["var"] = info["A_tiled_waves"], info["var_tiles"]
    eng.close()
    return out


def _short(names):
    """'spmv_tiled_kernel<1, 3> + spmv_long_partial_kernel<0> + ...' -> 'tiled<1, 3>+long'"""
    parts = [n.strip() for n in names.split(" + ")]
    out = [parts[0].replace("spmv_", "").replace("_kernel", "")]
    if len(parts) > 1:
        out.append("slabs" if "stream" in parts[1] or "sj" in parts[1] else "long")
    return "+".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--no-vendor", action="store_true")
    ap.add_argument("--env", default="", help="environment for the product library, e.g. 'PDHG_SPMV=stream'")
    args = ap.parse_args()
    for kv in args.env.split():
        k, v = kv.split("=")
        os.environ[k] = v
    from tools import vendor_spmv
    print("# product: HIP-event brackets over 30 trials (fused products: A xbar + dual step; A'y' + interaction sums); "
          "vendor: rocSPARSE CSR SpMV alone (y = A x), best of adaptive / rowsplit / lrb / nnzsplit, preprocessing apart; "
          "frac = algorithmic bytes of the fused product / time / 8 TB/s", flush=True)
    only = [s.strip() for s in args.only.split(",") if s.strip()]
    for title, kind, kw in SHAPES:
        sized = title.split("-")[0] in ("200K", "1M", "4M", "10M")
        # (a size-tagged shape is selected only by a token that names a size tag: "clustered" stays the 10M shape alone)
        if (only and not any(o in title and (not sized or any(t in o for t in ("200K-", "1M-", "4M-", "10M-"))) for o in only)) or \
                (not only and sized):
            continue
        p = make_shape(kind, **kw)
        r = product_ms(p)
        line = (f"{title:42s} dual {r['dual_ms']:.4f} ms ({r['dual_bytes'] / r['dual_ms'] / 8e9:.3f})  "
                f"aty {r['aty_ms']:.4f} ms ({r['aty_bytes'] / r['aty_ms'] / 8e9:.3f})  waves={r['waves']} var={r['var']}"
                f"  [{_short(r['kernels'][0])} | {_short(r['kernels'][1])}]")
        if not args.no_vendor:
            A = p.constraint_matrix.tocsr()
            va = vendor_spmv.time_csr(A)
            vt = vendor_spmv.time_csr(p.constraint_matrix.T.tocsr())
            line += (f"  | vendor A x {va.get('best_ms', float('nan')):.4f} ms ({va.get('best_alg')})  "
                     f"A'y {vt.get('best_ms', float('nan')):.4f} ms ({vt.get('best_alg')})  "
                     f"product/vendor {r['dual_ms'] / va['best_ms']:.2f} {r['aty_ms'] / vt['best_ms']:.2f}")
            line += f"  all: {({k: v for k, v in va.items() if k in vendor_spmv.ALGS})} {({k: v for k, v in vt.items() if k in vendor_spmv.ALGS})}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
