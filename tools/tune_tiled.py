#!/usr/bin/env python3
"""Dev tool: A/B the SpMV layouts/geometries on one generated problem, in one
process, interleaved and repeated (per-kernel HIP-event averages).
usage: tune_tiled.py [--m M --n N] "K=V K=V" "K=V" ..."""
import argparse, os, sys, time
os.environ.setdefault("PDHG_DEV", "1")     # development variables on (csrc/common.hpp: dev_env)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import folp_loader
pkg = folp_loader.load()
from firstorderlp_jl_amd.generators import random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_step
from firstorderlp_jl_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=10_000_000)
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--pagerank", type=int, default=0, help="PageRank LP with this many nodes instead of the random LP")
ap.add_argument("--colskew", action="store_true", help="random LP whose column popularity falls like 1/sqrt(index) (hub columns clustered at low indices, short rows)")
ap.add_argument("--banded", type=int, default=0, help="random LP whose row i has its k entries within +-BANDED columns of i*n/m (local / banded structure)")
ap.add_argument("--shape", default="", help="blockdiag | clustered | twodensity | arrowhead (structured test matrices, m x n, about k per row)")
ap.add_argument("cfgs", nargs="*")
a = ap.parse_args()
if a.pagerank:
    from firstorderlp_jl_amd.generators import pagerank_lp
    p = pagerank_lp(a.pagerank, seed=1)
elif a.shape:
    import scipy.sparse as sp
    from firstorderlp_jl_amd import linear_programming_problem
    rng = np.random.default_rng(7)
    rows = np.repeat(np.arange(a.m, dtype=np.int64), a.k)
    if a.shape == "blockdiag":            # 100 diagonal blocks, uniform inside a block
        nb = 100
        blk = rows * nb // a.m
        w = a.n // nb
        cols = blk * w + rng.integers(0, w, a.m * a.k)
    elif a.shape == "clustered":          # every row: k entries within +-500 columns of a random centre
        centre = np.repeat(rng.integers(0, a.n, a.m), a.k)
        cols = np.clip(centre + rng.integers(-500, 501, a.m * a.k), 0, a.n - 1)
    elif a.shape == "twodensity":         # alternating rows of k/3 and 5k/3 entries
        lens = np.where(np.arange(a.m) % 2 == 0, max(1, a.k // 3), 5 * a.k // 3)
        rows = np.repeat(np.arange(a.m, dtype=np.int64), lens)
        cols = rng.integers(0, a.n, rows.size)
    elif a.shape == "arrowhead":          # uniform + 5 dense rows + 5 dense columns
        cols = rng.integers(0, a.n, a.m * a.k)
        dr = np.repeat(np.arange(5, dtype=np.int64) * (a.m // 5), a.n // 4)
        dc = np.tile(rng.choice(a.n, a.n // 4, replace=False), 5)
        er = rng.choice(a.m, a.m // 4, replace=False)
        rows = np.concatenate([rows, dr, np.tile(er, 5)])
        cols = np.concatenate([cols, dc, np.repeat(np.arange(5, dtype=np.int64) * (a.n // 5) + 1, er.size)])
    else:
        raise SystemExit("unknown shape")
    M = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(a.m, a.n))
    M.sum_duplicates()
    p = linear_programming_problem(np.zeros(a.n), np.full(a.n, 10.0), rng.standard_normal(a.n), 0.0,
                                   M.tocsc(), rng.standard_normal(a.m), a.m // 2)
elif a.banded:
    import scipy.sparse as sp
    from firstorderlp_jl_amd import linear_programming_problem
    rng = np.random.default_rng(6)
    centre = (np.arange(a.m, dtype=np.int64) * a.n) // a.m
    cols = np.clip(np.repeat(centre, a.k) + rng.integers(-a.banded, a.banded + 1, a.m * a.k), 0, a.n - 1)
    rows = np.repeat(np.arange(a.m), a.k)
    M = sp.csr_matrix((rng.standard_normal(a.m * a.k), (rows, cols)), shape=(a.m, a.n))
    M.sum_duplicates()
    p = linear_programming_problem(np.zeros(a.n), np.full(a.n, 10.0), rng.standard_normal(a.n), 0.0,
                                   M.tocsc(), rng.standard_normal(a.m), a.m // 2)
elif a.colskew:
    import scipy.sparse as sp
    from firstorderlp_jl_amd import linear_programming_problem
    rng = np.random.default_rng(5)
    cols = np.minimum((rng.random(a.m * a.k) ** 2 * a.n).astype(np.int64), a.n - 1)     # density ~ 1/sqrt(index)
    rows = np.repeat(np.arange(a.m), a.k)
    M = sp.csr_matrix((rng.standard_normal(a.m * a.k), (rows, cols)), shape=(a.m, a.n))
    M.sum_duplicates()
    p = linear_programming_problem(np.zeros(a.n), np.full(a.n, 10.0), rng.standard_normal(a.n), 0.0,
                                   M.tocsc(), rng.standard_normal(a.m), a.m // 2)
else:
    p = random_lp(a.m, a.n, a.k, 12345)
A = p.constraint_matrix
step0 = 1.0 / float(np.abs(A.data).max())
pw0 = float(np.linalg.norm(p.objective_vector) / np.linalg.norm(p.right_hand_side))
KEYS = ["PDHG_SPMV", "PDHG_TILE_SHIFT", "PDHG_TILE_COLS", "PDHG_TILE_FILL", "PDHG_VAR_TILES", "PDHG_TW_NNZ_CAP", "PDHG_HOST_THREADS", "PDHG_TW_MIN_LDS_KB", "PDHG_TW_ROWS", "PDHG_TW_WPB", "PDHG_TW_FLAGS", "PDHG_XCD_REMAP",
        "PDHG_TW_MAX_ROWS", "PDHG_TW_WGS_PER_CU", "PDHG_SLABS", "PDHG_SLAB_MB", "PDHG_GRAPH"]
for rep in range(a.reps):
    for cfg in a.cfgs or [""]:
        for k in KEYS:
            os.environ.pop(k, None)
        for kv in cfg.split():
            k, v = kv.split("=")
            os.environ[k] = v
        t0 = time.time()
        eng = pkg.HipPdhgEngine.from_problem(p)
        tc = time.time() - t0
        st = PdhgSolverState(eng, step_size=step0, primal_weight=pw0)
        for _ in range(5):
            take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
        eng.profile_enable(True)
        for _ in range(a.steps):
            take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
        c1, m1 = eng.profile_read(_lib.K_SPMV_DUAL)
        c2, m2 = eng.profile_read(_lib.K_SPMV_ATY)
        by1 = eng.kernel_algorithmic_bytes(_lib.K_SPMV_DUAL)
        info = eng.layout_info()
        print(f"rep{rep} [{cfg:48s}] dual {m1/c1:.4f} ms ({by1/(m1/c1)/1e6:6.0f} GB/s)  aty {m2/c2:.4f} ms  "
              f"waves={info['A_tiled_waves']} var={info['var_tiles']} create={tc:.1f}s", flush=True)
        eng.close()
