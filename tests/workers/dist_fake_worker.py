#!/usr/bin/env python3
"""One rank of a world-N group whose ranks ALL use one GPU, over the test-only RCCL stand-in
(tests/fake_rccl; the parent sets PDHG_RCCL_LIB).  Spawned by tests/test_gpu_fake_rccl.py through
``python -m torch.distributed.run``.  Exit code 0 = every check passed on every rank.

argv: traj <ingest: global|rows> <overlap 0|1> <lp: small|tiled>
        the trajectory of tests/workers/dist_rank_worker.py through pdhg_create_dist / pdhg_create_dist_rows; every rank
        must hold the same bits; rank 0 also runs (a) the IN-PROCESS shard group on the same device (peer-kernel back
        end: the same rank-ordered sums, so NOT A BIT may differ) and (b) the single handle (decisions equal, 1e-9)
      kat <case names, comma-separated>
        the reference's known-answer tests (tests/kat_common.py <- test/test_primal_dual_hybrid_gradient.jl:77-423)
        with every rank driving its shard: optimize() + device rescaling + device evaluation over the group
      optimize
        solve_qp.jl's defaults on a random LP, group vs single handle
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import folp_loader  # noqa: E402

pkg = folp_loader.load()
from firstorderlp_jl_amd import HipPdhgEngine  # noqa: E402
from firstorderlp_jl_amd.distributed import (make_row_partitioned_hip_engine,  # noqa: E402
                                             make_row_shard_hip_engine, row_partitioned_factory, row_shard_of)
from firstorderlp_jl_amd.generators import random_lp  # noqa: E402
from tests.workers.dist_rank_worker import run  # noqa: E402

DEVICE = 0


def finish(rc, eng=None):
    flag = torch.tensor([rc])
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    dist.barrier()
    if eng is not None:
        eng.close()
    dist.destroy_process_group()
    sys.exit(int(flag.item()))


def random_shape_lp(lp):
    """lp = 'rand:<seed>': a random LP whose shape comes from the seed -- 1 ... 12, 300 or 4 000 rows and columns (fewer rows than ranks,
    fewer columns than ranks x chunks included), 1 ... 8 entries per row."""
    seed = int(lp[5:])
    rng = np.random.default_rng(seed)
    top = int(rng.choice([12, 300, 4000]))       # (300: slices of a few 16-column lines -- ragged last chunks)
    m = int(rng.integers(1, top))
    n = int(rng.integers(1, top))
    return random_lp(m, n, int(rng.integers(1, min(8, n) + 1)), seed=seed), 25, 8


def traj(ingest, overlap, lp):
    os.environ["PDHG_DIST_OVERLAP"] = overlap
    rank, world = dist.get_rank(), dist.get_world_size()
    if lp.startswith("rand:"):
        p, steps, mp_steps = random_shape_lp(lp)
    elif lp == "tiled":
        os.environ["PDHG_DIST_ROUND_WGS"] = "64"
        p, steps, mp_steps = random_lp(1_100_000, 600_000, 5, seed=21), 30, 8
    else:
        p, steps, mp_steps = random_lp(30000, 20000, 6, seed=21), 60, 25
    if ingest == "rows":
        bounds = HipPdhgEngine.partition_rows(p.constraint_matrix, world)
        eng = make_row_shard_hip_engine(row_shard_of(p, bounds, rank), device_id=DEVICE)
    else:
        eng = make_row_partitioned_hip_engine(p, device_id=DEVICE)
    info = eng.dist_info()
    assert info["world"] == world and info["rank"] == rank and info["backend"] == 0 and info["local_ranks"] == 1, info
    assert "fake_rccl" in HipPdhgEngine.rccl_info()["path"], HipPdhgEngine.rccl_info()
    g = run(eng, p, steps, mp_steps)
    mine = np.concatenate([g["x"], g["y"], g["xa"], g["ya"], g["xm"], g["ym"], g["aty"], g["ax"],
                           [g["step"], g["mp_step"], float(sum(g["decisions"]))]])
    ref = torch.from_numpy(mine.copy())
    dist.broadcast(ref, src=0)
    rc = 0
    if not np.array_equal(ref.numpy(), mine, equal_nan=True):
        print(f"rank {rank} diverged from rank 0", flush=True)
        rc = 1
    if rank == 0:
        try:
            # (a) the in-process group: same shards, same rank-ordered sums -> bitwise
            geng = HipPdhgEngine.from_problem(p, device_ids=[DEVICE] * world)
            assert geng.dist_info()["backend"] == 1, ("in-process group", geng.dist_info())
            inproc = run(geng, p, steps, mp_steps)
            geng.close()
            for k, v in g.items():
                assert np.array_equal(np.asarray(v), np.asarray(inproc[k]), equal_nan=True), f"{k}: processes over the stand-in != in-process group"
            # (b) the single handle
            s = run(HipPdhgEngine.from_problem(p, device_id=DEVICE), p, steps, mp_steps)
            A = p.constraint_matrix
            assert g["decisions"] == s["decisions"], "accept/reject decisions differ from the single handle"
            assert g["step"] == s["step"] or abs(g["step"] - s["step"]) <= 1e-9 * s["step"], ("step size", g["step"], s["step"])
            for k in ("x", "y", "xa", "ya", "xm", "ym"):
                np.testing.assert_allclose(g[k], s[k], rtol=1e-9, atol=1e-9, err_msg=k)
            assert g["mp_iters"] == s["mp_iters"], ("Malitsky-Pock iterations", g["mp_iters"], s["mp_iters"])
            if not np.isnan(g["ya"]).any():      # (nothing averaged: no restart to the average took place, tests/workers/dist_rank_worker.py)
                np.testing.assert_allclose(g["aty"], A.T @ g["ya"], rtol=1e-11, atol=1e-11)
            np.testing.assert_allclose(g["ax"], A @ g["x"], rtol=1e-11, atol=1e-11)
            print(f"fake worker ok: traj world {world} ingest {ingest} overlap {overlap} lp {lp}, {sum(g['decisions'])} trials, "
                  f"layout {eng.layout_info().get('At_tiled_waves')}", flush=True)
        except AssertionError as exc:
            print("fake worker FAILED: " + " | ".join(str(exc).split("\n")[:8]), flush=True)
            rc = 1
    finish(rc, eng)


def agtraj(ingest, lp):
    """The all-gather of xbar overlapped with A_p xbar (PDHG_DIST_AG_OVERLAP=1: xbar in column chunks on the comm stream, one
    product pass per chunk) against the SAME passes behind one all-gather (=2): not a bit may differ, on any rank; rank 0
    also runs the in-process group (peer back end: the passes behind its all-gather -> bitwise too), the group without
    chunks and the single handle (another order of additions inside a row: decisions equal, iterates to 1e-9)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if lp.startswith("rand:"):
        p, steps, mp_steps = random_shape_lp(lp)
    elif lp == "tiled":
        os.environ["PDHG_SPMV"] = "tiled"
        p, steps, mp_steps = random_lp(300_000, 200_000, 6, seed=22), 30, 8
    else:
        p, steps, mp_steps = random_lp(30000, 20000, 6, seed=21), 60, 25
    bounds = HipPdhgEngine.partition_rows(p.constraint_matrix, world)

    def make():
        if ingest == "rows":
            return make_row_shard_hip_engine(row_shard_of(p, bounds, rank), device_id=DEVICE)
        return make_row_partitioned_hip_engine(p, device_id=DEVICE)

    def cat(g):
        return np.concatenate([g["x"], g["y"], g["xa"], g["ya"], g["xm"], g["ym"], g["aty"], g["ax"],
                               [g["step"], g["mp_step"], float(sum(g["decisions"]))]])
    rc = 0
    runs = {}
    for mode in ("1", "2"):
        os.environ["PDHG_DIST_AG_OVERLAP"] = mode
        eng = make()
        d = eng.layout_describe()
        want = "overlapped with A_p xbar" if mode == "1" else "passes behind one all-gather"
        chunked = "all_gather" in d      # (a slice too short to cut -- fewer columns per rank than chunks -- keeps the one all-gather)
        assert chunked or lp.startswith("rand:"), d
        if chunked:
            assert d["all_gather"]["chunks"] >= 2 and d["all_gather"]["mode"] == want, d.get("all_gather")
        runs[mode] = run(eng, p, steps, mp_steps)
        layouts = [c["layout"] for c in d["all_gather"]["passes"]] if chunked else []
        eng.close()
        dist.barrier()
    a, b = cat(runs["1"]), cat(runs["2"])
    if not np.array_equal(a, b, equal_nan=True):
        print(f"rank {rank}: the overlapped all-gather differs from the passes behind one all-gather", flush=True)
        rc = 1
    ref = torch.from_numpy(a.copy())
    dist.broadcast(ref, src=0)
    if not np.array_equal(ref.numpy(), a, equal_nan=True):
        print(f"rank {rank} diverged from rank 0", flush=True)
        rc = 1
    if rank == 0:
        try:
            g = runs["1"]
            os.environ["PDHG_DIST_AG_OVERLAP"] = "1"
            geng = HipPdhgEngine.from_problem(p, device_ids=[DEVICE] * world)
            assert geng.dist_info()["backend"] == 1, ("in-process group", geng.dist_info())
            assert not chunked or geng.layout_describe()["all_gather"]["mode"] == "passes behind one all-gather"
            inproc = run(geng, p, steps, mp_steps)
            geng.close()
            for k, v in g.items():
                assert np.array_equal(np.asarray(v), np.asarray(inproc[k]), equal_nan=True), f"{k}: processes over the stand-in != in-process group"
            os.environ.pop("PDHG_DIST_AG_OVERLAP")
            plain = HipPdhgEngine.from_problem(p, device_ids=[DEVICE] * world)
            assert "all_gather" not in plain.layout_describe()
            s1 = run(plain, p, steps, mp_steps)
            plain.close()
            s2 = run(HipPdhgEngine.from_problem(p, device_id=DEVICE), p, steps, mp_steps)
            for s in (s1, s2):
                assert g["decisions"] == s["decisions"], "accept/reject decisions differ"
                assert g["step"] == s["step"] or abs(g["step"] - s["step"]) <= 1e-9 * s["step"], ("step size", g["step"], s["step"])
                for k in ("x", "y", "xa", "ya", "xm", "ym"):
                    np.testing.assert_allclose(g[k], s[k], rtol=1e-9, atol=1e-9, err_msg=k)
            print(f"fake worker ok: agtraj world {world} ingest {ingest} lp {lp}, {sum(g['decisions'])} trials, pass layouts {layouts}", flush=True)
        except AssertionError as exc:
            print("fake worker FAILED: " + " | ".join(str(exc).split("\n")[:8]), flush=True)
            rc = 1
    os.environ.pop("PDHG_DIST_AG_OVERLAP", None)
    finish(rc)


def kat(names):
    from tests import kat_common
    rank, world = dist.get_rank(), dist.get_world_size()
    cases = {c.__name__: c for c in kat_common.CASES}
    rc = 0
    made = []

    def factory(problem):
        eng = row_partitioned_factory(device_id=DEVICE)(problem)
        assert eng.dist_info()["world"] == world and eng.dist_info()["backend"] == 0
        made.append(eng)
        return eng
    factory.takes_original_problem = True
    for name in names.split(","):
        try:
            cases[name](factory)
            if rank == 0:
                print(f"fake worker ok: kat {name} world {world}", flush=True)
        except AssertionError as exc:
            print(f"fake worker FAILED: kat {name} rank {rank}: {exc}", flush=True)
            rc = 1
            break
    finish(rc)


def whole_solve():
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import optimize
    from tests.test_gpu_end_to_end import _params
    rank, world = dist.get_rank(), dist.get_world_size()
    tol = 1e-6
    p = random_lp(12000, 10000, 8, seed=42)
    grp = optimize(_params(tol, 40000), p, row_partitioned_factory(device_id=DEVICE))
    rc = 0
    # every rank took the same decisions
    mine = torch.tensor([float(grp.iteration_count), grp.iteration_stats[-1].convergence_information[0].primal_objective])
    ref = mine.clone()
    dist.broadcast(ref, src=0)
    if not torch.equal(ref, mine):
        print(f"rank {rank}: iteration count / objective differ from rank 0: {mine} vs {ref}", flush=True)
        rc = 1
    if rank == 0:
        try:
            one = optimize(_params(tol, 40000), p)
            assert one.termination_string == grp.termination_string == "OPTIMAL", (one.termination_string, grp.termination_string)
            c1 = one.iteration_stats[-1].convergence_information[0]
            c2 = grp.iteration_stats[-1].convergence_information[0]
            scale = 1.0 + abs(c1.primal_objective)
            assert abs(c2.primal_objective - c1.primal_objective) <= 50 * tol * scale
            assert abs(c2.dual_objective - c1.dual_objective) <= 50 * tol * scale
            assert 0.5 <= grp.iteration_count / one.iteration_count <= 2.0
            print(f"fake worker ok: optimize world {world}: {grp.iteration_count} iterations (single handle {one.iteration_count})", flush=True)
        except AssertionError as exc:
            print(f"fake worker FAILED: optimize: {exc}", flush=True)
            rc = 1
    finish(rc)


def main():
    dist.init_process_group("gloo")
    mode = sys.argv[1]
    if mode == "traj":
        traj(*sys.argv[2:5])
    elif mode == "agtraj":
        agtraj(*sys.argv[2:4])
    elif mode == "kat":
        kat(sys.argv[2])
    else:
        whole_solve()


if __name__ == "__main__":
    main()
