"""The row-partitioned (multi-GPU) host path on CPU: 2 processes, gloo, with
the CPU oracle injected as each rank's local engine.  Checks that the sharded
run reproduces the unsharded oracle: same accept/reject decisions, iterates
equal to 1e-12 (the only differences are the 2-term rank-ordered sums of the
reduce-scatter and the per-slice numpy dots).  The engine under test is the
numpy mirror of csrc/dist.hpp (tests/dist_mirror.py: RowPartitionedEngine): row shards,
owned column slices, reduce-scatter -> slice -> all-gather, scalars combined in
rank order."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import folp_loader
    folp_loader.load()
    import torch.distributed as dist
    from firstorderlp_jl_amd.distributed import partition_rows, shard_rows
    from tests.dist_mirror import RowPartitionedEngine, TorchComm
    from firstorderlp_jl_amd.generators import random_lp
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (
        AdaptiveStepsizeParams, PdhgSolverState, take_step)
    from tests.oracle_engine import OracleEngine
    from tests import helpers as H
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = random_lp(600, 500, 6, seed=5)
        # the row partition comes from the PRODUCT library (pdhg_partition_rows: host-only C code of csrc/dist.hpp, no GPU
        # needed), evaluated independently on every rank; the package's numpy statement of the same rule must agree
        from firstorderlp_jl_amd import HipPdhgEngine
        bounds = HipPdhgEngine.partition_rows(p.constraint_matrix, world)
        ranges = [(int(bounds[k]), int(bounds[k + 1])) for k in range(world)]
        assert ranges == partition_rows(p.constraint_matrix, world)
        lo, hi = ranges[rank]
        local = OracleEngine(**shard_rows(p, lo, hi))
        eng = RowPartitionedEngine(local, TorchComm(), ranges)
        step, pw = H.initial_step_and_weight(p)
        state = PdhgSolverState(eng, step_size=step, primal_weight=pw)
        for _ in range(40):
            take_step(AdaptiveStepsizeParams(0.3, 0.6), state)
        x, y = eng.get_current()
        xa, ya = eng.get_average()
        # restart to the average exercises the A'y refresh all-reduce
        eng.restart_to_average()
        aty = eng.get_dual_product()
        for _ in range(5):
            take_step(AdaptiveStepsizeParams(0.3, 0.6), state)
        x2, y2 = eng.get_current()
        q.put((rank, ranges, x, y, xa, ya, aty, x2, y2, state.step_size,
               state.total_number_iterations))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_row_partition_matches_unsharded_oracle():
    sys.path.insert(0, ROOT)
    from firstorderlp_jl_amd.generators import random_lp
    from tests.oracle_engine import OracleEngine
    from tests import helpers as H
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (
        AdaptiveStepsizeParams, PdhgSolverState, take_step)
    world = 2
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0

    p = random_lp(600, 500, 6, seed=5)
    ref = OracleEngine.from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    state = PdhgSolverState(ref, step_size=step, primal_weight=pw)
    for _ in range(40):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), state)
    x, y = ref.get_current()
    xa, ya = ref.get_average()
    ref.restart_to_average()
    aty = ref.get_dual_product()
    for _ in range(5):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), state)
    x2, y2 = ref.get_current()

    ranges = results[0][1]
    assert ranges[0][0] == 0 and ranges[-1][1] == 600 and ranges[0][1] == ranges[1][0]
    for (rank, _, rx, ry, rxa, rya, raty, rx2, ry2, rstep, rtot) in results:
        assert rtot == state.total_number_iterations
        assert abs(rstep - state.step_size) <= 1e-12 * state.step_size
        for got, want in ((rx, x), (ry, y), (rxa, xa), (rya, ya), (raty, aty), (rx2, x2), (ry2, y2)):
            np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-11)
    # replicas are bitwise identical across ranks
    for a, b in zip(results[0][2:9], results[1][2:9]):
        assert np.array_equal(a, b)


def test_partition_rows_balances_nnz():
    sys.path.insert(0, ROOT)
    from firstorderlp_jl_amd.distributed import partition_rows, shard_rows
    from tests import helpers as H
    p = H.skewed_lp(400, 300, seed=3, dense_rows=2, dense_cols=1)
    for world in (1, 2, 3, 8):
        ranges = partition_rows(p.constraint_matrix, world)
        assert len(ranges) == world and ranges[0][0] == 0 and ranges[-1][1] == 400
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        tot = 0
        for lo, hi in ranges:
            sh = shard_rows(p, lo, hi)
            tot += sh["constraint_matrix"].nnz
            assert sh["num_equalities"] == min(max(p.num_equalities - lo, 0), hi - lo)
        assert tot == p.constraint_matrix.nnz


def _kat_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import folp_loader
    folp_loader.load()
    import torch.distributed as dist
    from firstorderlp_jl_amd.distributed import partition_rows, shard_rows
    from tests.dist_mirror import RowPartitionedEngine, TorchComm
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import optimize
    from firstorderlp_jl_amd.saddle_point import RestartScheme
    from tests import kat_common
    from tests.oracle_engine import OracleEngine
    from tests import helpers as H
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def factory(problem):
            ranges = partition_rows(problem.constraint_matrix, world)
            lo, hi = ranges[rank]
            return RowPartitionedEngine(OracleEngine(**shard_rows(problem, lo, hi)), TorchComm(), ranges)
        params = kat_common.generate_primal_dual_hybrid_gradient_params(
            iteration_limit=600, restart_scheme=RestartScheme.ADAPTIVE_NORMALIZED,
            l_inf_ruiz_iterations=3)
        out = optimize(params, H.example_lp(), factory)
        q.put((rank, out.primal_solution, out.dual_solution, out.iteration_count,
               out.termination_reason.name))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_full_optimize_row_partitioned_two_ranks():
    """A whole reference-style solve (rescaling + adaptive restarts + termination
    evaluation through the sharded A*x / A'*y) on 2 ranks: both ranks return the
    reference KAT's optimum (test_primal_dual_hybrid_gradient.jl:130-147)."""
    world = 2
    port = 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_kat_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    # the same solve, unsharded, on the oracle engine
    sys.path.insert(0, ROOT)
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import optimize
    from firstorderlp_jl_amd.saddle_point import RestartScheme
    from tests import kat_common
    from tests.oracle_engine import OracleEngine
    from tests import helpers as H
    params = kat_common.generate_primal_dual_hybrid_gradient_params(
        iteration_limit=600, restart_scheme=RestartScheme.ADAPTIVE_NORMALIZED,
        l_inf_ruiz_iterations=3)
    ref = optimize(params, H.example_lp(), OracleEngine.from_problem)
    for (rank, x, y, iters, reason) in results:
        # restart decisions are discrete, so sharded and unsharded runs may take
        # different (equally valid) paths; both must land on the LP's optimum
        np.testing.assert_allclose(x, [1.0, 0.0, 6.0, 2.0], atol=1e-9)
        np.testing.assert_allclose(y, [0.5, 4.0, 0.0], atol=1e-9)
        np.testing.assert_allclose(x, ref.primal_solution, atol=1e-9)
        np.testing.assert_allclose(y, ref.dual_solution, atol=1e-9)
        # exact convergence makes movement == 0 -> NUMERICAL_ERROR exit (pdhg.jl:691-695)
        assert iters <= 600 and reason in ("TERMINATION_REASON_ITERATION_LIMIT",
                                           "TERMINATION_REASON_NUMERICAL_ERROR")
    assert np.array_equal(results[0][1], results[1][1]) and np.array_equal(results[0][2], results[1][2])


# ---- Malitsky-Pock linesearch in the row-partitioned form ---------------------

MP_STEPS = 30


def _mp_params():
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import MalitskyPockStepsizeParameters
    return MalitskyPockStepsizeParameters(downscaling_factor=0.7, breaking_factor=0.99,
                                          interpolation_coefficient=1.0)


def _mp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import folp_loader
    folp_loader.load()
    import torch.distributed as dist
    from firstorderlp_jl_amd.distributed import partition_rows, shard_rows
    from tests.dist_mirror import RowPartitionedEngine, TorchComm
    from firstorderlp_jl_amd.generators import random_lp
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import PdhgSolverState, take_step
    from tests.oracle_engine import OracleEngine
    from tests import helpers as H
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = random_lp(500, 400, 5, seed=11)
        ranges = partition_rows(p.constraint_matrix, world)
        lo, hi = ranges[rank]
        eng = RowPartitionedEngine(OracleEngine(**shard_rows(p, lo, hi)), TorchComm(), ranges)
        step, pw = H.initial_step_and_weight(p)
        state = PdhgSolverState(eng, step_size=step, primal_weight=pw, ratio_step_sizes=1.0)
        for _ in range(MP_STEPS):
            take_step(_mp_params(), state)
        x, y = eng.get_current()
        xa, ya = eng.get_average()
        q.put((rank, x, y, xa, ya, state.step_size, state.ratio_step_sizes,
               state.total_number_iterations, eng.average_info()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_malitsky_pock_row_partitioned_matches_unsharded_oracle():
    sys.path.insert(0, ROOT)
    from firstorderlp_jl_amd.generators import random_lp
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import PdhgSolverState, take_step
    from tests.oracle_engine import OracleEngine
    from tests import helpers as H
    world = 2
    port = 27500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mp_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    p = random_lp(500, 400, 5, seed=11)
    ref = OracleEngine.from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    state = PdhgSolverState(ref, step_size=step, primal_weight=pw, ratio_step_sizes=1.0)
    for _ in range(MP_STEPS):
        take_step(_mp_params(), state)
    x, y = ref.get_current()
    xa, ya = ref.get_average()
    for (rank, rx, ry, rxa, rya, rstep, rratio, rtot, rinfo) in results:
        assert rtot == state.total_number_iterations     # same linesearch trip counts
        assert abs(rstep - state.step_size) <= 1e-12 * state.step_size
        assert abs(rratio - state.ratio_step_sizes) <= 1e-12
        assert rinfo[0] == ref.average_info()[0]
        for got, want in ((rx, x), (ry, y), (rxa, xa), (rya, ya)):
            np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-11)


# ---- QP (objective matrix replicated on every rank) in the row-partitioned form ----

def _qp_problem():
    import scipy.sparse as sp
    from firstorderlp_jl_amd.generators import random_lp
    p = random_lp(300, 200, 5, seed=23)
    rng = np.random.default_rng(4)
    B = sp.random(200, 200, density=0.02, random_state=5, format="csc")
    p.objective_matrix = sp.csc_matrix(B.T @ B + sp.diags(rng.uniform(0.0, 1.0, 200)))
    return p


def _qp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import folp_loader
    folp_loader.load()
    import torch.distributed as dist
    from firstorderlp_jl_amd.distributed import partition_rows, shard_rows
    from tests.dist_mirror import RowPartitionedEngine, TorchComm
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (
        AdaptiveStepsizeParams, PdhgSolverState, take_step)
    from tests.oracle_engine import OracleEngine
    from tests import helpers as H
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = _qp_problem()
        ranges = partition_rows(p.constraint_matrix, world)
        lo, hi = ranges[rank]
        eng = RowPartitionedEngine(OracleEngine(**shard_rows(p, lo, hi)), TorchComm(), ranges)
        step, pw = H.initial_step_and_weight(p)
        state = PdhgSolverState(eng, step_size=step, primal_weight=pw)
        for _ in range(40):
            take_step(AdaptiveStepsizeParams(0.3, 0.6), state)
        x, y = eng.get_current()
        q.put((rank, x, y, state.step_size, state.total_number_iterations))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_qp_row_partitioned_matches_unsharded_oracle():
    sys.path.insert(0, ROOT)
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (
        AdaptiveStepsizeParams, PdhgSolverState, take_step)
    from tests.oracle_engine import OracleEngine
    from tests import helpers as H
    world = 2
    port = 25500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_qp_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    p = _qp_problem()
    ref = OracleEngine.from_problem(p)
    step, pw = H.initial_step_and_weight(p)
    state = PdhgSolverState(ref, step_size=step, primal_weight=pw)
    for _ in range(40):
        take_step(AdaptiveStepsizeParams(0.3, 0.6), state)
    x, y = ref.get_current()
    for (rank, rx, ry, rstep, rtot) in results:
        assert rtot == state.total_number_iterations
        assert abs(rstep - state.step_size) <= 1e-12 * state.step_size
        np.testing.assert_allclose(rx, x, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(ry, y, rtol=1e-11, atol=1e-11)


# ---- bench.py's N > 1 harness: generate once, cut with the library's partition, one slice per rank ----

def _bench_shard_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import argparse
    import folp_loader
    folp_loader.load()
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        args = argparse.Namespace(m=3000, n=2500, nnz_per_row=7, seed=11, pagerank_nodes=4000)
        ctx = {"dist": dist, "rank": rank, "world": world}
        shard, meta = bench.shard_for_rank(args, "random", ctx)
        q.put((rank, meta, shard["row_bounds"].tolist(), shard["constraint_rows"].tocsr(), shard["right_hand_side_rows"],
               shard["objective_vector"], shard["num_equalities"], shard["m_global"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_bench_generates_once_and_every_rank_gets_only_its_rows(world):
    """`bench.py --gpus N` (what the driver's scaling run launches): rank 0 generates the LP, cuts it with
    pdhg_partition_rows (host-only) and ships one slice per rank through files; the slices must tile
    the matrix exactly and carry the global vectors.  Runs on CPU: nothing here needs a GPU."""
    import scipy.sparse as sp
    sys.path.insert(0, ROOT)
    from firstorderlp_jl_amd.generators import random_lp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29730 + world
    procs = [ctx.Process(target=_bench_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    p = random_lp(3000, 2500, 7, seed=11)
    A = p.constraint_matrix.tocsr()
    bounds = got[0][2]
    assert bounds[0] == 0 and bounds[-1] == 3000 and len(bounds) == world + 1
    for rank, meta, b, rows, rhs, c, num_eq, m_global in got:
        assert b == bounds and m_global == 3000 and num_eq == p.num_equalities
        assert meta["m"] == 3000 and meta["n"] == 2500 and meta["nnz"] == A.nnz
        lo, hi = bounds[rank], bounds[rank + 1]
        assert rows.shape == (hi - lo, 2500)
        assert (rows != A[lo:hi]).nnz == 0
        assert np.array_equal(rhs, p.right_hand_side[lo:hi]) and np.array_equal(c, p.objective_vector)
    stacked = sp.vstack([g[3] for g in got]).tocsr()
    assert (stacked != A).nnz == 0
    # nnz-balanced: no rank holds more than its share plus one row's worth
    per_rank = [g[3].nnz for g in got]
    assert max(per_rank) <= A.nnz / world + A.getnnz(axis=1).max() + 1
