#!/usr/bin/env python3
"""Random LARGE shapes through the library's own dispatch (no layout forced) against the CPU oracle: the builder's
automatic rules -- sliced jagged copies, column slabs, the L2-tiled sweep, long-row chunks -- only engage beyond ~1 000 row
blocks, where tests/test_gpu_property.py's shapes never go.  Per example: the layout picked, A x and A'y
(tests/helpers.assert_products_match_oracle: short rows bitwise, long rows 1e-13 * sum |a x|), one trial step and five
adaptive take_steps against the oracle's loop.  Usage: python tools/big_shape_hunt.py [examples] [seed]"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PDHG_DEV", "1")
import folp_loader  # noqa: E402
folp_loader.load()      # registers the package directory under its importable name
from firstorderlp_jl_amd import HipPdhgEngine, linear_programming_problem  # noqa: E402
from tests import helpers as H  # noqa: E402


def make(rng):
    law = rng.choice(["uniform", "banded", "blockdiag", "powerlaw", "hubs", "fixed_t"])
    m = int(rng.integers(150_000, 2_500_000))
    n = int(rng.integers(150_000, 2_500_000))
    nnz = int(rng.integers(2_000_000, 12_000_000))
    k = max(1, nnz // m)
    if law == "fixed_t":       # columns of one length: the transpose has the fixed-length rows
        cols = np.repeat(np.arange(n), max(1, nnz // n))
        rows = rng.integers(0, m, cols.size)
    else:
        lens = np.full(m, k) if law in ("banded", "blockdiag") else rng.poisson(k, m)
        if law == "hubs":
            hub = rng.choice(m, 6, replace=False)
            lens[hub] = rng.integers(3_000, min(n, 400_000), 6)
        rows = np.repeat(np.arange(m), lens)
        if law == "banded":
            w = int(rng.integers(50, 60_000))
            cols = (rows * (n / m)).astype(np.int64) + rng.integers(-w, w + 1, rows.size)
            cols = np.clip(cols, 0, n - 1)
        elif law == "blockdiag":
            nb = int(rng.integers(8, 4000))
            blk = rows * nb // m
            lo = blk * n // nb
            cols = lo + (rng.random(rows.size) * (np.maximum(1, (blk + 1) * n // nb - lo))).astype(np.int64)
        elif law == "powerlaw":
            cols = np.minimum(n - 1, (n * rng.random(rows.size) ** 3).astype(np.int64))
        else:
            cols = rng.integers(0, n, rows.size)
            if law == "hubs":      # and a few dense columns
                extra_r = rng.integers(0, m, 300_000)
                rows = np.concatenate([rows, extra_r]); cols = np.concatenate([cols, rng.integers(0, 3, extra_r.size)])
    A = sp.csc_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(m, n))
    A.sort_indices()
    lb = np.where(rng.random(n) < 0.25, -np.inf, rng.integers(-2, 1, n).astype(float))
    ub = np.where(rng.random(n) < 0.4, np.inf, np.maximum(lb, 0.0) + rng.integers(0, 3, n))
    p = linear_programming_problem(lb, ub, rng.standard_normal(n), 0.0, A, rng.standard_normal(m), int(rng.integers(0, m)))
    return law, p


def brief(desc):
    out = {}
    for k in ("A", "At"):
        d = desc.get(k, {})
        out[k] = d.get("kernel", d.get("form", "?")) if isinstance(d, dict) else "?"
        if isinstance(d, dict) and d.get("sliced_jagged"):
            out[k] = f"sj(G={d['sliced_jagged']['slices_per_wave']}, hubs={d['sliced_jagged']['hub_rows']})"
    return out


def main():
    examples = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    failures = 0
    for ex in range(examples):
        t0 = time.time()
        law, p = make(rng)
        A = p.constraint_matrix
        m, n = A.shape
        eng = HipPdhgEngine.from_problem(p)
        oracle = H.oracle_from_problem(p)
        li = eng.layout_info()
        tag = (f"[{ex}] {law} {m}x{n} nnz={A.nnz} A: blocks={li['A_blocks']} sj={li['A_sj']} slabs={li['A_slabs']} "
               f"tiled={int(li['A_tiled_waves'] > 0)} long={li['A_long_rows']} | At: blocks={li['At_blocks']} sj={li['At_sj']} "
               f"slabs={li['At_slabs']} tiled={int(li['At_tiled_waves'] > 0)} long={li['At_long_rows']}")
        try:
            x, y = rng.standard_normal(n), rng.standard_normal(m)
            H.assert_products_match_oracle(eng, A, x, y, label=law)
            absA = abs(A).tocsr()
            eng.set_current(x, y)
            oracle.x, oracle.y = x.copy(), y.copy()
            oracle.recompute_dual_product()
            oracle.aty = eng.get_dual_product().copy()
            eng.trial_step(0.05, 2.0, 1.0)
            _, xn, yn, an = oracle.trial_step(0.05, 2.0, 1.0)
            gx, gy, ga = eng.get_trial()
            assert np.array_equal(gx, xn), "x' differs"
            tol = 1e-13 * (0.1 * (absA @ np.abs(2 * xn - x) + np.abs(p.right_hand_side)) + np.abs(y))
            assert np.all(np.abs(gy - yn) <= tol + 1e-300), "y' beyond the bound"
            assert np.all(np.abs(ga - an) <= 1e-12 * (absA.T @ np.abs(yn)) + 1e-300), "A'y' beyond the bound"
            # five adaptive steps from the origin
            eng.set_current(np.zeros(n), np.zeros(m))
            oracle.x, oracle.y = np.zeros(n), np.zeros(m)
            oracle.recompute_dual_product()
            step, pw = H.initial_step_and_weight(p)
            oracle.step_size, oracle.primal_weight, oracle.ratio_step_sizes = step, pw, 1.0
            it0 = oracle.total_number_iterations
            step, it, kkt, err, done = eng.take_steps_adaptive(5, 0.3, 0.6, step, pw, 0, 0.0)
            for _ in range(5):
                oracle.take_step_adaptive(0.3, 0.6)
            assert it == oracle.total_number_iterations - it0, ("trials", it, oracle.total_number_iterations - it0)
            assert abs(step - oracle.step_size) <= 1e-9 * step, "step size"
            xe, ye = eng.get_current()
            assert np.allclose(xe, oracle.x, rtol=1e-9, atol=1e-9 * np.abs(oracle.x).max()), "x after 5 steps"
            assert np.allclose(ye, oracle.y, rtol=1e-9, atol=1e-9 * np.abs(oracle.y).max()), "y after 5 steps"
            print(tag, f"ok {time.time() - t0:.1f}s", flush=True)
        except AssertionError as e:
            failures += 1
            print(tag, "FAILED:", e, flush=True)
        finally:
            eng.close()
            oracle.close()
    print(f"{examples} examples, {failures} failures (seed {seed})")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
