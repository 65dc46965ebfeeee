import os, sys
sys.path.insert(0, os.getcwd())
import folp_loader; pkg = folp_loader.load()
import torch
from firstorderlp_jl_amd.generators import random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_steps
from tests import helpers as H
def used():
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return (t - f) / 2**20
probs = [random_lp(4000, 3500, 8, 5), random_lp(60000, 50000, 8, 6)]
base = None
for it in range(40):
    for p in probs:
        for dev in ([None], [0, 0]):
            eng = pkg.HipPdhgEngine.from_problem(p) if dev == [None] else pkg.HipPdhgEngine.from_problem(p, device_ids=dev)
            step, pw = H.initial_step_and_weight(p)
            st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
            take_steps(AdaptiveStepsizeParams(0.3, 0.6), st, 30)
            eng.save_restart_point()
            eng.trust_region_bound(0, 1.0, 1.0, 0.5, 0, False)
            eng.close()
    if it == 2: base = used()
    if it in (2, 10, 20, 39): print(f"iteration {it}: device memory in use {used():.1f} MiB", flush=True)
print("growth since iteration 2: %.1f MiB" % (used() - base))
