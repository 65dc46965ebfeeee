#!/usr/bin/env python3
"""dev: the sliced jagged kernel's three epilogue modes on one shape, for a rocprofv3 --kernel-trace --stats run:
plain products of A and A' (pdhg_spmv / pdhg_spmv_t) beside the fused ones of take_step."""
import os
import sys
os.environ.setdefault("PDHG_DEV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.shape_table import SHAPES, make_shape
import folp_loader
pkg = folp_loader.load()
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_step

want = sys.argv[1] if len(sys.argv) > 1 else "blockdiag"
title, kind, kw = [s for s in SHAPES if want in s[0]][0]
p = make_shape(kind, **kw)
eng = pkg.HipPdhgEngine.from_problem(p)
A = p.constraint_matrix
st = PdhgSolverState(eng, step_size=1.0 / float(np.abs(A.data).max()), primal_weight=1.0)
rng = np.random.default_rng(0)
x, y = rng.standard_normal(A.shape[1]), rng.standard_normal(A.shape[0])
for _ in range(6):
    eng.spmv(x)
    eng.spmv_t(y)
for _ in range(12):
    take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
print(title, eng.layout_info())
