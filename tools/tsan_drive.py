#!/usr/bin/env python3
"""Drives the shard pool (csrc/dist.hpp: one issuing host thread per shard, peer-kernel back end, several shards on ONE
GPU) through libpdhg_hip_tsan.so.  Run by tools/archive/r4_tsan_shards.sh with the TSan runtime preloaded; prints TSAN_DRIVE_OK
when the run itself worked -- ThreadSanitizer's reports, if any, go to stderr / the log file."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import folp_loader  # noqa: E402

pkg = folp_loader.load()
from firstorderlp_jl_amd.generators import random_lp  # noqa: E402
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_step, take_steps  # noqa: E402

shards = int(sys.argv[1]) if len(sys.argv) > 1 else 4
p = random_lp(60000, 50000, 8, seed=3)
eng = pkg.HipPdhgEngine.from_problem(p, device_ids=[0] * shards)
step = 1.0 / np.abs(p.constraint_matrix.data).max()
st = PdhgSolverState(eng, step_size=step, primal_weight=1.0)
pol = AdaptiveStepsizeParams(0.3, 0.6)
for _ in range(20):
    take_step(pol, st)
take_steps(pol, st, 60)
x, y = eng.get_current()
xa, ya = eng.get_average()
eng.close()
print("TSAN_DRIVE_OK", shards, st.total_number_iterations, float(np.abs(x).sum()), float(np.abs(ya).sum()))
