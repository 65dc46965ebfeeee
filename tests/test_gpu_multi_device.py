"""The library-owned row-partitioned form on MORE THAN ONE physical GPU: RCCL with world > 1.

Everything here skips (cleanly, with the reason) on a box with one GPU and turns itself on
the moment ``pytest -m gpu`` runs where >= 2 are visible.  What it proves then is what the
1-GPU box cannot: ``ncclCommInitAll`` on distinct devices with one issuing host thread per
shard, ``ncclReduceScatter`` / ``ncclAllGather`` / ``ncclReduce`` / ``ncclBroadcast`` between
devices, the one-process-per-GPU route with both ingest forms, and that every result equals
the single handle's (arithmetic being distributed: src/primal_dual_hybrid_gradient.jl:442-549).
The same checks run on ONE GPU through the peer-kernel back end and 1-rank communicators in
tests/test_gpu_dist_group.py.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.distributed import multi_device_factory
from firstorderlp_jl_amd.generators import random_lp
from tests import helpers as H
from tests import kat_common
from tests.test_gpu_dist_group import _compare, _run

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_count():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


needs2 = pytest.mark.skipif(device_count() < 2, reason=f"needs >= 2 GPUs, {device_count()} visible")
needs4 = pytest.mark.skipif(device_count() < 4, reason=f"needs >= 4 GPUs, {device_count()} visible")


def _spawn(world, ingest, overlap="0", timeout=540, ag="0"):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + world),
           os.path.join(ROOT, "tests", "workers", "dist_rank_worker.py"), ingest, "rank", overlap, ag]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "dist worker ok" in r.stdout, r.stdout[-2000:]


@needs2
@pytest.mark.parametrize("threads", ["1", "0"], ids=["thread_per_shard", "single_thread_issue"])
@pytest.mark.parametrize("overlap", ["0", "1"], ids=["reduce_scatter", "per_slice_reduce"])
def test_create_multi_on_two_gpus_matches_single_handle(gpu_required, monkeypatch, overlap, threads):
    monkeypatch.setenv("PDHG_DIST_OVERLAP", overlap)
    monkeypatch.setenv("PDHG_SHARD_THREADS", threads)
    p = random_lp(30000, 20000, 6, seed=21)
    geng = HipPdhgEngine.from_problem(p, device_ids=[0, 1])
    info = geng.dist_info()
    assert info["world"] == info["local_ranks"] == 2 and info["backend"] == 0     # RCCL, not peer kernels
    g = _run(geng, p, 60, 25)
    s = _run(HipPdhgEngine.from_problem(p, device_id=0), p, 60, 25)
    _compare(g, s, p)
    trials, issue, wait = geng.host_issue_stats()
    assert trials > 0 and issue > 0


@needs2
def test_create_multi_peer_kernels_across_two_gpus(gpu_required, monkeypatch):
    monkeypatch.setenv("PDHG_COMM", "p2p")            # hipDeviceEnablePeerAccess + kernels reading the peer's buffers
    p = random_lp(30000, 20000, 6, seed=21)
    geng = HipPdhgEngine.from_problem(p, device_ids=[0, 1])
    assert geng.dist_info()["backend"] == 1
    _compare(_run(geng, p, 60, 25), _run(HipPdhgEngine.from_problem(p, device_id=0), p, 60, 25), p)


@needs4
def test_create_multi_on_four_gpus_tiled_shards(gpu_required):
    p = random_lp(2_200_000, 1_200_000, 5, seed=21)   # shards large enough for the tiled sweep
    geng = HipPdhgEngine.from_problem(p, device_ids=[0, 1, 2, 3])
    assert geng.layout_info()["A_tiled_waves"] > 0
    _compare(_run(geng, p, 40, 10), _run(HipPdhgEngine.from_problem(p, device_id=0), p, 40, 10), p)


@needs2
@pytest.mark.parametrize("ingest", ["global", "rows"])
@pytest.mark.parametrize("overlap", ["0", "1"])
def test_one_process_per_gpu_two_ranks(gpu_required, ingest, overlap):
    """pdhg_create_dist / pdhg_create_dist_rows in two spawned processes, rank r on GPU r."""
    _spawn(2, ingest, overlap)


@needs4
def test_one_process_per_gpu_four_ranks_rank_local_ingest(gpu_required):
    _spawn(4, "rows")


_KATS = [c for c in kat_common.CASES
         if c.__name__ in ("low_precision", "high_precision", "adaptive_restart_heuristic", "malitsky_pock_no_smoothing",
                           "quadratic_programming_1", "ruiz", "l2_norm_rescaling", "lp_without_bounds",
                           "correlation_clustering_triangle_plus")]


@needs2
@pytest.mark.parametrize("threads", ["1", "0"], ids=["thread_per_shard", "single_thread_issue"])
@pytest.mark.parametrize("case", _KATS, ids=lambda c: c.__name__)
def test_reference_kats_on_two_gpus(gpu_required, monkeypatch, case, threads):
    """Some of the reference's known-answer tests (test/test_primal_dual_hybrid_gradient.jl:77-423)
    through a two-GPU RCCL group: device evaluation, rescaling, restarts, a QP, Malitsky-Pock."""
    monkeypatch.setenv("PDHG_SHARD_THREADS", threads)

    def factory(problem):
        eng = HipPdhgEngine.from_problem(problem, device_ids=[0, 1])
        assert eng.dist_info()["backend"] == 0
        return eng
    case(factory)


@needs2
def test_optimize_with_device_evaluation_and_rescaling_on_two_gpus(gpu_required):
    """optimize() with solve_qp.jl's defaults, everything on the devices: two GPUs vs one handle."""
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import optimize
    from tests.test_gpu_end_to_end import _params
    tol = 1e-6
    p = random_lp(12000, 10000, 8, seed=42)
    one = optimize(_params(tol, 40000), p)
    grp = optimize(_params(tol, 40000), p, multi_device_factory([0, 1]))
    assert one.termination_string == grp.termination_string == "OPTIMAL"
    c1 = one.iteration_stats[-1].convergence_information[0]
    c2 = grp.iteration_stats[-1].convergence_information[0]
    scale = 1.0 + abs(c1.primal_objective)
    assert abs(c2.primal_objective - c1.primal_objective) <= 50 * tol * scale
    assert abs(c2.dual_objective - c1.dual_objective) <= 50 * tol * scale
    assert 0.5 <= grp.iteration_count / one.iteration_count <= 2.0


# ---- round 4: the persistent group trial kernels (csrc/group_kernel.hpp) ACROSS devices ---------------------------------
# On one device they are the default and bitwise the per-launch path (tests/test_gpu_dist_group.py).  Across devices the
# protocol -- peer stores of xbar, system-scope write-back / invalidate around cross-device flag barriers, the owner's
# rank-ordered reduce of the peers' partials -- has never executed, so it is opt-in (PDHG_GROUP_COOP=1) and this is the
# test that decides whether it may become the default: not a bit may differ from the per-launch path.

@needs2
@pytest.mark.parametrize("device_ids", [[0, 1], [0, 1, 0, 1]], ids=["one_shard_per_gpu", "two_shards_per_gpu"])
def test_group_trial_kernels_across_two_gpus_are_bitwise_the_per_launch_path(gpu_required, monkeypatch, device_ids):
    monkeypatch.setenv("PDHG_COMM", "p2p")             # the peer back end (the kernels exchange through peer-mapped memory)
    p = random_lp(30000, 20000, 6, seed=21)
    runs = {}
    for coop in ("1", "0"):
        monkeypatch.setenv("PDHG_GROUP_COOP", coop)
        eng = HipPdhgEngine.from_problem(p, device_ids=device_ids)
        assert eng.dist_info()["backend"] == 1
        runs[coop] = _run(eng, p, 60, 25)
        runs[coop]["trials"] = eng.layout_info()["group_coop_trials"]
        eng.close()
    assert runs["1"].pop("trials") >= 60 and runs["0"].pop("trials") == 0
    for key, val in runs["1"].items():
        assert np.array_equal(np.asarray(val), np.asarray(runs["0"][key])), key


# ---- round 6: the all-gather of xbar in column chunks beside A_p xbar (PDHG_DIST_AG_OVERLAP; csrc/dist.hpp) on real devices:
# RCCL's broadcasts on the comm stream while the product passes run.  Over the test transport (one GPU) this is
# tests/test_gpu_fake_rccl.py::test_all_gather_overlapped_with_the_product_is_bitwise_the_passes_behind_one_all_gather.
@needs2
@pytest.mark.parametrize("ag", ["1", "2"], ids=["overlapped", "passes_behind_one_all_gather"])
@pytest.mark.parametrize("overlap", ["0", "1"], ids=["reduce_scatter", "per_slice_reduce"])
def test_one_process_per_gpu_two_ranks_all_gather_in_column_chunks(gpu_required, ag, overlap):
    _spawn(2, "rows", overlap, ag=ag)


@needs2
@pytest.mark.parametrize("threads", ["1", "0"], ids=["thread_per_shard", "single_thread_issue"])
def test_create_multi_on_two_gpus_all_gather_overlap_is_bitwise_the_passes_behind_one_all_gather(gpu_required, monkeypatch, threads):
    monkeypatch.setenv("PDHG_SHARD_THREADS", threads)
    p = random_lp(30000, 20000, 6, seed=21)
    runs = {}
    for mode in ("1", "2"):
        monkeypatch.setenv("PDHG_DIST_AG_OVERLAP", mode)
        geng = HipPdhgEngine.from_problem(p, device_ids=[0, 1])
        assert geng.dist_info()["backend"] == 0 and geng.layout_describe()["all_gather"]["chunks"] >= 2
        runs[mode] = _run(geng, p, 40, 10)
        geng.close()
    for k, v in runs["1"].items():
        assert np.array_equal(np.asarray(v), np.asarray(runs["2"][k])), k
    monkeypatch.delenv("PDHG_DIST_AG_OVERLAP")
    _compare(runs["1"], _run(HipPdhgEngine.from_problem(p), p, 40, 10), p)


@needs4
def test_one_process_per_gpu_four_ranks_all_gather_in_column_chunks(gpu_required):
    _spawn(4, "rows", "1", ag="1")
