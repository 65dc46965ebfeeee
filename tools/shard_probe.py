#!/usr/bin/env python3
"""Dev tool: per-rank compute of the row-partitioned form at the real shard
shapes of config S, measured on ONE GPU (rank 0's shard for P = 2, 4, 8; the
all-reduce is skipped, so iterates are not meaningful -- only kernel times)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import folp_loader
pkg = folp_loader.load()
from firstorderlp_jl_amd.generators import random_lp
from firstorderlp_jl_amd.distributed import partition_rows, shard_rows
from firstorderlp_jl_amd import _lib

p = random_lp(10_000_000, 10_000_000, 10, 12345)
step0 = 1.0 / float(np.abs(p.constraint_matrix.data).max())
for P in [int(v) for v in os.environ.get("SHARD_P", "1,2,4,8").split(",")]:
    lo, hi = partition_rows(p.constraint_matrix, P)[0]
    t0 = time.time()
    eng = pkg.HipPdhgEngine(**shard_rows(p, lo, hi))
    tc = time.time() - t0
    for _ in range(3):
        eng.dist_trial_begin(step0, 1.0, 1.0); eng.dist_trial_end(); eng.accept(step0)
    eng.profile_enable(True)
    t0 = time.perf_counter()
    N = 20
    for _ in range(N):
        eng.dist_trial_begin(step0, 1.0, 1.0); eng.dist_trial_end(); eng.accept(step0)
    wall = (time.perf_counter() - t0) / N * 1e3
    ks = {eng.kernel_name(k).split("<")[0][-22:] + ("" if k not in (1, 2) else ("_A" if k == 1 else "_At")):
          round(eng.profile_read(k)[1] / max(eng.profile_read(k)[0], 1), 4) for k in range(_lib.K_COUNT)}
    eng.profile_enable(False)
    t0 = time.perf_counter()
    for _ in range(N):
        eng.dist_trial_begin(step0, 1.0, 1.0); eng.dist_trial_end(); eng.accept(step0)
    wall2 = (time.perf_counter() - t0) / N * 1e3
    print(f"P={P} rows[{lo},{hi}) create={tc:.1f}s compute-only iteration {wall2:.3f} ms (profiled {wall:.3f})", ks,
          {k: v for k, v in eng.layout_info().items() if "tiled" in k or "shift" in k}, flush=True)
    eng.close()
