"""The one-process-per-GPU routes of csrc/dist.hpp EXECUTED with world 2 / 4 / 8 on a ONE-GPU box.

Real RCCL refuses two ranks on one device, so until round 5 ``pdhg_create_dist`` / ``pdhg_create_dist_rows`` had only
ever run with world = 1.  Here every rank is a real process (``python -m torch.distributed.run``), all on cuda:0, and the
library's run-time RCCL binding (csrc/rccl_loader.hpp, ``PDHG_RCCL_LIB``) is pointed at the test-only stand-in
``tests/fake_rccl`` (host-staged exchange through a shared mapping, rank-ordered sums).  What runs is the PRODUCT code:
rank-local ingest, the all-gather of xbar, the reduce-scatter / per-slice reduce of A_p'y_p on the comm stream,
``combine_scalars`` over ncclAllGather, the row-range broadcasts of the getters, device evaluation, rescaling, restarts.
Checks: every rank holds the same bits; the multi-process result is BITWISE the in-process shard group's (same
rank-ordered sums); decisions equal the single handle's and iterates agree to 1e-9; the reference's KATs pass; a whole
``optimize`` terminates OPTIMAL like the single handle.  (Arithmetic being distributed:
src/primal_dual_hybrid_gradient.jl:442-549, 653-731.)"""
import os
import subprocess
import sys

import pytest

from tests import fake_rccl

pytestmark = [pytest.mark.gpu, pytest.mark.own_row_order]     # rows of <= 256 entries: the row order changes nothing here
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PORT = [29700]


def _spawn(world, *argv, timeout=840, **env_extra):
    _PORT[0] += 1
    env = fake_rccl.env(FAKE_RCCL_TIMEOUT_S=240, **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_PORT[0]),
           os.path.join(ROOT, "tests", "workers", "dist_fake_worker.py"), *argv]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "fake worker ok" in r.stdout and "FAILED" not in r.stdout, r.stdout[-2000:]
    return r.stdout


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,ingest,overlap", [(w, i, o) for w in (2, 4) for i in ("global", "rows") for o in ("0", "1")] +
                         [(8, "global", "0"), (8, "rows", "1")],       # (eight processes sharing one GPU take ~20 s per case)
                         ids=lambda v: {"0": "reduce_scatter", "1": "per_slice_reduce"}.get(v, str(v)))
def test_processes_on_one_gpu_match_the_in_process_group_bitwise(gpu_required, world, ingest, overlap):
    _spawn(world, "traj", ingest, overlap, "small")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("overlap", ["0", "1"], ids=["reduce_scatter", "per_slice_reduce"])
def test_two_processes_tiled_shards_with_the_product_cut_into_rounds(gpu_required, overlap):
    """A_p' tiled and launched a residency round at a time, slice k reduced (ncclReduce on the comm stream) while the
    next round computes: the overlapped exchange between real processes."""
    out = _spawn(2, "traj", "rows", overlap, "tiled")
    assert "layout 0" not in out


_KATS = ("low_precision,high_precision,adaptive_restart_heuristic,malitsky_pock_no_smoothing,quadratic_programming_1,"
         "ruiz,l2_norm_rescaling,lp_without_bounds,correlation_clustering_triangle_plus")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3])
def test_reference_kats_between_processes(gpu_required, world):
    _spawn(world, "kat", _KATS)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4])
def test_optimize_with_device_evaluation_and_rescaling_between_processes(gpu_required, world):
    _spawn(world, "optimize")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,ingest,lp", [(2, "global", "small"), (4, "rows", "small"), (2, "rows", "tiled")])     # (world 8: `bench.py --gpus 8 --dist-overlap` over the
                         # same transport, profiles/r06_bench_fake_rccl_8ranks_overlap.json; the eight-process case took 100 s here)
def test_all_gather_overlapped_with_the_product_is_bitwise_the_passes_behind_one_all_gather(gpu_required, world, ingest, lp):
    """VERDICT r5 #3 / SURVEY 8e(ii): xbar travels in column chunks on the comm stream (one ncclAllGather per chunk into a
    chunk-major copy), A_p xbar runs as one pass per chunk (carried row sums), pass c waiting for chunk c only.  Bitwise the same
    passes behind one all-gather, on every rank; bitwise the in-process group; decisions of the unchunked group and of the
    single handle, iterates to 1e-9 (pdhg.jl:472-494)."""
    out = _spawn(world, "agtraj", ingest, lp)
    if lp == "tiled":
        assert "sweep" in out, out[-600:]


@pytest.mark.timeout(900)
def test_reference_kats_with_the_overlapped_all_gather(gpu_required):
    _spawn(2, "kat", _KATS, PDHG_DIST_AG_OVERLAP="1")


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,mode,ingest,seed", [(3, "traj1", "global", 1180834645),      # 2 x 1: fewer rows than ranks, nothing averaged
                                                     (3, "agtraj", "rows", 1413644620),       # 6 x 1: a slice too short to cut into chunks
                                                     (4, "agtraj", "rows", 544083118),
                                                     (2, "traj0", "rows", 2037209174)])
def test_random_shapes_between_processes(gpu_required, world, mode, ingest, seed):
    """Four cases of tools/dist_shape_hunt.py (random LP shapes through the one-process-per-GPU routes; 330 cases in the
    round's hunts, profiles/r06_dist_shape_hunt.txt), among them the degenerate shapes its checker first tripped over."""
    if mode == "agtraj":
        _spawn(world, "agtraj", ingest, f"rand:{seed}", PDHG_DEV="1", PDHG_DIST_AG_CHUNKS="3")
    else:
        _spawn(world, "traj", ingest, mode[-1], f"rand:{seed}")
