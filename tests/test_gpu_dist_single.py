"""The row-partitioned HIP path end to end on ONE GPU: RCCL communicator of
world_size 1 (backend "nccl"), exchange buffer wrapped zero-copy as a torch
tensor, all-reduce on the engine's stream.  (Multi-GPU boxes are only
available to the driver; the N>1 logic is covered by tests/test_distributed_gloo.py.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("parts", ["1", "4"])
def test_row_partitioned_engine_world1_matches_single_engine(gpu_required, parts, monkeypatch):
    # parts=4: the exchange is issued range by range with async RCCL all-reduces
    # (pdhg_dist_trial_begin_part), overlapping the partial products of A_p'y'_p
    monkeypatch.setenv("PDHG_DIST_PARTS", parts)
    monkeypatch.setenv("PDHG_DIST_ROUND_WGS", "64")   # parts are whole residency rounds (512 WGs) by default
    import torch
    import torch.distributed as dist
    from firstorderlp_jl_amd import HipPdhgEngine
    from firstorderlp_jl_amd.distributed import make_row_partitioned_hip_engine
    from firstorderlp_jl_amd.generators import random_lp
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (
        AdaptiveStepsizeParams, PdhgSolverState, take_step)
    from tests import helpers as H

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29700 + 2 * (os.getpid() % 200) + (parts == "4"))
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        created = True
    try:
        p = random_lp(700000, 600000, 6, seed=9) if parts == "4" else random_lp(20000, 15000, 8, seed=9)
        deng = make_row_partitioned_hip_engine(p, device_id=0)
        if parts == "4":   # large enough for the tiled layout, which is what can be cut into parts
            assert len(deng._parts()) == 5
        seng = HipPdhgEngine.from_problem(p)
        step, pw = H.initial_step_and_weight(p)
        ds = PdhgSolverState(deng, step_size=step, primal_weight=pw)
        ss = PdhgSolverState(seng, step_size=step, primal_weight=pw)
        for _ in range(30):
            take_step(AdaptiveStepsizeParams(0.3, 0.6), ds)
            take_step(AdaptiveStepsizeParams(0.3, 0.6), ss)
        assert ds.total_number_iterations == ss.total_number_iterations
        xd, yd = deng.get_current()
        xs, ys = seng.get_current()
        # identical SpMV kernels; only the dx/dAty reduction kernel differs
        np.testing.assert_allclose(xd, xs, rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(yd, ys, rtol=1e-10, atol=1e-10)
        deng.restart_to_average()
        seng.restart_to_average()
        np.testing.assert_allclose(deng.get_dual_product(), seng.get_dual_product(),
                                   rtol=1e-10, atol=1e-10)
        y_new = np.abs(np.random.default_rng(0).standard_normal(deng.m))
        deng.set_current(None, y_new)
        seng.set_current(None, y_new)
        assert np.array_equal(deng.get_dual_product(), seng.get_dual_product())
    finally:
        if created:
            dist.destroy_process_group()
