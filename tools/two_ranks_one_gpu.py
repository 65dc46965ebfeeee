#!/usr/bin/env python3
"""Dev probe: TWO processes, each one rank of a pdhg_create_dist group, BOTH on GPU 0 --
does this RCCL build accept two ranks of one communicator on the same device?  If it does,
the multi-process route (real collectives between processes) can be exercised on a 1-GPU box.
launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
        --master-port 29517 tools/two_ranks_one_gpu.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch.distributed as dist
import folp_loader
pkg = folp_loader.load()
from firstorderlp_jl_amd.distributed import make_row_partitioned_hip_engine
from firstorderlp_jl_amd.generators import random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import AdaptiveStepsizeParams, PdhgSolverState, take_step
from tests import helpers as H

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
p = random_lp(6000, 5003, 8, seed=11)
try:
    eng = make_row_partitioned_hip_engine(p, device_id=0)
except Exception as exc:
    print(f"rank {rank}: pdhg_create_dist on a shared device failed: {exc!r}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if "Duplicate" in repr(exc) or "invalid" in repr(exc).lower() else 3)
step, pw = H.initial_step_and_weight(p)
st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
for _ in range(40):
    take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
x, y = eng.get_current()
xa, ya = eng.get_average()
if rank == 0:
    single = pkg.HipPdhgEngine.from_problem(p, device_id=0)
    s1 = PdhgSolverState(single, step_size=step, primal_weight=pw)
    for _ in range(40):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), s1)
    x1, y1 = single.get_current()
    xa1, ya1 = single.get_average()
    err = max(np.abs(x - x1).max(), np.abs(y - y1).max(), np.abs(xa - xa1).max(), np.abs(ya - ya1).max())
    print(f"two ranks on one GPU: {st.total_number_iterations} trials (single engine {s1.total_number_iterations}), "
          f"max |difference| to the single engine {err:.3e}", flush=True)
    assert st.total_number_iterations == s1.total_number_iterations and err < 1e-9
dist.barrier()
eng.close()
dist.destroy_process_group()
