"""Where did tests/test_gpu_sj.py's QP case spend 133 s at n = 60 000?  Times generation, create, rescale, trials (dev aid)."""
import os, sys, time
os.environ.setdefault("PDHG_DEV", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
import folp_loader; folp_loader.load()
from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_step
from tests import helpers as H
n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 60_000, 40_000
t = time.time(); p = random_lp(m, n, 6, seed=31); rng = np.random.default_rng(4)
B = sp.random(n, n, density=3.0 / n, random_state=9, format="csc")
p.objective_matrix = sp.csc_matrix(B.T @ B + sp.diags(rng.uniform(0.5, 3.0, n))); p.objective_matrix.sum_duplicates(); p.objective_matrix.sort_indices()
print("generate", round(time.time() - t, 2), "Q nnz", p.objective_matrix.nnz, flush=True)
for sj in ("1", "0"):
    os.environ.update(PDHG_SPMV="stream", PDHG_COOP="0", PDHG_GRAPH="1", PDHG_SJ=sj, PDHG_SLABS="0")
    t = time.time(); e = HipPdhgEngine.from_problem(p); print("sj", sj, "create", round(time.time() - t, 2), e.layout_describe().get("Q", {}).get("layout"), flush=True)
    t = time.time(); e.rescale(10, False, 1.0); print("  rescale", round(time.time() - t, 2), flush=True)
    step, pw = H.initial_step_and_weight(p); st = PdhgSolverState(e, step_size=step, primal_weight=pw)
    t = time.time()
    for _ in range(30): take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
    print("  30 steps", round(time.time() - t, 2), st.total_number_iterations, flush=True)
