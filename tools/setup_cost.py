#!/usr/bin/env python3
"""What happens before the first iteration: pdhg_create (ingest, layouts, the policies settled by timing), device
rescaling, the first batch of take_steps (kernel code loading, graph / persistent-kernel set-up).
Measured round 6 (MI355X): L1-SVM 30.6 / 5.8 / 2.1 ms (create / rescale / first 40 steps; the next 40: 1.7), PageRank-1M
69 / 43 / 19.7 (8.2), random 1M 33 / 16 / 7.1 (6.9).
Usage: python tools/setup_cost.py [l1svm|pagerank|<n>]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import folp_loader
pkg = folp_loader.load()
from firstorderlp_jl_amd.generators import random_lp, l1_svm_rcv1_like_lp, pagerank_lp

arg = sys.argv[1] if len(sys.argv) > 1 else "l1svm"
p = l1_svm_rcv1_like_lp() if arg == "l1svm" else (pagerank_lp(1_000_000) if arg == "pagerank" else random_lp(int(arg), int(arg), 10, 12345))
import torch
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()      # the HIP context itself is not counted
def clock(name, f):
    t0 = time.perf_counter(); r = f(); dt = time.perf_counter() - t0
    print(f"  {name:58s} {dt * 1e3:9.2f} ms", flush=True)
    return r
print(arg, f"m={p.num_constraints} n={p.num_variables} nnz={p.constraint_matrix.nnz}")
for rnd in (1, 2):
    print(f" handle {rnd} of this process")
    eng = clock("pdhg_create (HipPdhgEngine.from_problem)", lambda: pkg.HipPdhgEngine.from_problem(p))
    clock("pdhg_rescale (Ruiz 10 + Pock-Chambolle 1.0)", lambda: eng.rescale(10, False, 1.0))
    step = 1.0 / eng.matrix_max_abs()
    st = [step, 0, 0.0]
    def steps(n):
        s, it, kkt, err, done = eng.take_steps_adaptive(n, 0.3, 0.6, st[0], 1.0, st[1], st[2]); st[:] = [s, it, kkt]
    clock("first 40 take_steps", lambda: steps(40))
    clock("next 40 take_steps", lambda: steps(40))
    clock("close", eng.close)
