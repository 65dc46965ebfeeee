#!/usr/bin/env python3
"""Dev tool: time the kernels of pdhg_trial_step on the config-S random LP with
whatever library PDHG_HIP_LIB points at (tools/variants.sh builds -D variants).
The same trial is repeated from a fixed iterate, so diagnostic variants that
compute wrong values cannot derail the step-size logic."""
import argparse, os, sys
os.environ.setdefault("PDHG_DEV", "1")     # development variables on (csrc/common.hpp: dev_env)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import folp_loader
pkg = folp_loader.load()
from firstorderlp_jl_amd.generators import random_lp
from firstorderlp_jl_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=10_000_000)
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("libs", nargs="*", help="library paths (default: the in-tree build)")
a = ap.parse_args()
p = random_lp(a.m, a.n, a.k, 12345)
step = 1.0 / float(np.abs(p.constraint_matrix.data).max())
for rep in range(a.reps):
    for path in a.libs or [_lib.LIB_PATH]:
        _lib._lib, _lib.LIB_PATH = None, os.path.abspath(path)   # one process, several builds
        eng = pkg.HipPdhgEngine.from_problem(p)
        for _ in range(5):
            eng.trial_step(step, 1.0)
        eng.profile_enable(True)
        for _ in range(a.steps):
            eng.trial_step(step, 1.0)
        c1, m1 = eng.profile_read(_lib.K_SPMV_DUAL)
        c2, m2 = eng.profile_read(_lib.K_SPMV_ATY)
        print(f"rep{rep} [{os.path.basename(path):32s}] dual {m1/c1:.4f} ms  aty {m2/c2:.4f} ms", flush=True)
        eng.close()
