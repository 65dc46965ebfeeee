#!/usr/bin/env python3
"""One rank of a pdhg_create_dist / pdhg_create_dist_rows group, run as its own process
(tests/test_gpu_multi_device.py and tests/test_gpu_dist_group.py spawn it through
``python -m torch.distributed.run``).  Every rank runs the same trajectory on its shard;
rank 0 also runs the single-handle engine and compares.  Exit code 0 = all checks passed.

argv: <ingest: global|rows> <devices: 'rank' (rank r on GPU r) | int (every rank on that GPU)>
      [overlap 0|1] [all-gather overlap 0|1|2: PDHG_DIST_AG_OVERLAP]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

import folp_loader  # noqa: E402

pkg = folp_loader.load()
from firstorderlp_jl_amd import HipPdhgEngine  # noqa: E402
from firstorderlp_jl_amd.distributed import (make_row_partitioned_hip_engine,  # noqa: E402
                                             make_row_shard_hip_engine, row_shard_of)
from firstorderlp_jl_amd.generators import random_lp  # noqa: E402
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (  # noqa: E402
    AdaptiveStepsizeParams, MalitskyPockStepsizeParameters, PdhgSolverState, take_step)
from tests import helpers as H  # noqa: E402


def run(eng, p, steps=60, mp_steps=25):
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    decisions = []
    for _ in range(steps):
        before = st.total_number_iterations
        take_step(AdaptiveStepsizeParams(0.3, 0.6), st)
        decisions.append(st.total_number_iterations - before)
    out = dict(decisions=decisions, step=st.step_size)
    out["x"], out["y"] = eng.get_current()
    out["xa"], out["ya"] = eng.get_average()
    if eng.average_info()[0] > 0:       # (a degenerate random LP can stop at its first trial: zero movement, nothing averaged)
        eng.restart_to_average()
    out["aty"] = eng.get_dual_product()
    out["ax"] = eng.spmv(out["x"])
    ms = PdhgSolverState(eng, step_size=st.step_size, primal_weight=pw, ratio_step_sizes=1.0)
    mp = MalitskyPockStepsizeParameters(downscaling_factor=0.7, breaking_factor=0.99, interpolation_coefficient=1.0)
    for _ in range(mp_steps):
        take_step(mp, ms)
    out["xm"], out["ym"] = eng.get_current()
    out["mp_iters"], out["mp_step"] = ms.total_number_iterations, ms.step_size
    return out


def main():
    ingest, devices = sys.argv[1], sys.argv[2]
    if len(sys.argv) > 3:
        os.environ["PDHG_DIST_OVERLAP"] = sys.argv[3]
    if len(sys.argv) > 4 and sys.argv[4] != "0":
        os.environ["PDHG_DIST_AG_OVERLAP"] = sys.argv[4]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    device = rank if devices == "rank" else int(devices)
    p = random_lp(30000, 20000, 6, seed=21)          # every rank can afford the small LP; only `global` ingests all of it
    if ingest == "rows":
        bounds = HipPdhgEngine.partition_rows(p.constraint_matrix, world)
        eng = make_row_shard_hip_engine(row_shard_of(p, bounds, rank), device_id=device)
    else:
        eng = make_row_partitioned_hip_engine(p, device_id=device)
    info = eng.dist_info()
    assert info["world"] == world and info["rank"] == rank and info["backend"] == 0, info
    g = run(eng, p)
    # all ranks must hold the same bits (scalars are combined in rank order on every rank)
    mine = np.concatenate([g["x"], g["y"], g["xa"], g["ya"], [g["step"], float(sum(g["decisions"]))]])
    import torch
    ref = torch.from_numpy(mine.copy())
    dist.broadcast(ref, src=0)
    assert np.array_equal(ref.numpy(), mine), f"rank {rank} diverged from rank 0"
    rc = 0
    if rank == 0:
        s = run(HipPdhgEngine.from_problem(p, device_id=device), p)
        A = p.constraint_matrix
        try:
            assert g["decisions"] == s["decisions"], "accept/reject decisions differ"
            assert abs(g["step"] - s["step"]) <= 1e-9 * s["step"]
            for k in ("x", "y", "xa", "ya", "xm", "ym"):
                np.testing.assert_allclose(g[k], s[k], rtol=1e-9, atol=1e-9, err_msg=k)
            assert g["mp_iters"] == s["mp_iters"]
            np.testing.assert_allclose(g["aty"], A.T @ g["ya"], rtol=1e-11, atol=1e-11)
            np.testing.assert_allclose(g["ax"], A @ g["x"], rtol=1e-11, atol=1e-11)
            print(f"dist worker ok: world {world}, ingest {ingest}, {sum(g['decisions'])} trials, "
                  f"rccl {HipPdhgEngine.rccl_info()}", flush=True)
        except AssertionError as exc:
            print(f"dist worker FAILED: {exc}", flush=True)
            rc = 1
    flag = torch.tensor([rc])
    dist.broadcast(flag, src=0)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()
    sys.exit(int(flag.item()))


if __name__ == "__main__":
    main()
