"""The library-owned row-partitioned form (csrc/dist.hpp) on the ONE GPU of the
test box.

* several shards of one LP inside one process, all on cuda:0 (``device_ids=[0, 0]``,
  ``[0, 0, 0]``): the exchange then runs through the peer-kernel back end -- the
  complete reduce-scatter -> slice -> all-gather pipeline, rank-ordered sums, scalar
  combination -- and is compared with the single-handle engine and scipy;
* RCCL itself with one rank: ``device_ids=[0]`` (ncclCommInitAll) and
  ``unique_id/rank/world`` (ncclCommInitRank), i.e. real ncclAllGather /
  ncclReduceScatter / ncclBroadcast calls from the library (RCCL refuses two ranks
  on one device, and multi-GPU boxes are only available to the driver);
* the reference's 22 PDHG KATs (test/test_primal_dual_hybrid_gradient.jl:77-423)
  through a two-shard and a three-shard group, stream and tiled layouts;
* device evaluation, trust-region bounds, restarts and rescaling on a group.
The world_size > 1 host logic on CPU is tests/test_distributed_gloo.py."""
import os

import numpy as np
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from firstorderlp_jl_amd.generators import random_lp
from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (
    AdaptiveStepsizeParams, MalitskyPockStepsizeParameters, PdhgSolverState, take_step)
from tests import helpers as H
from tests import kat_common

pytestmark = pytest.mark.gpu

ADAPTIVE = AdaptiveStepsizeParams(0.3, 0.6)


def MP_PARAMS():
    return MalitskyPockStepsizeParameters(downscaling_factor=0.7, breaking_factor=0.99,
                                          interpolation_coefficient=1.0)


def _run(eng, p, steps, mp_steps=0):
    step, pw = H.initial_step_and_weight(p)
    st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
    decisions = []
    for _ in range(steps):
        before = st.total_number_iterations
        take_step(ADAPTIVE, st)
        decisions.append(st.total_number_iterations - before)
    out = dict(decisions=decisions, step=st.step_size)
    out["x"], out["y"] = eng.get_current()
    out["xa"], out["ya"] = eng.get_average()
    eng.restart_to_average()
    out["aty"] = eng.get_dual_product()
    out["ax"] = eng.spmv(out["x"])
    if mp_steps:
        ms = PdhgSolverState(eng, step_size=st.step_size, primal_weight=pw, ratio_step_sizes=1.0)
        for _ in range(mp_steps):
            take_step(MP_PARAMS(), ms)
        out["xm"], out["ym"] = eng.get_current()
        out["mp_iters"], out["mp_step"] = ms.total_number_iterations, ms.step_size
    return out


def _compare(g, s, p, tol=1e-9):
    assert g["decisions"] == s["decisions"]
    assert abs(g["step"] - s["step"]) <= 1e-9 * s["step"]
    for k in ("x", "y", "xa", "ya"):
        np.testing.assert_allclose(g[k], s[k], rtol=tol, atol=tol)
    A = p.constraint_matrix
    # A'y_avg refreshed after the restart (sharded dual product) and the sharded A*x, vs scipy
    np.testing.assert_allclose(g["aty"], A.T @ g["ya"], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(g["ax"], A @ g["x"], rtol=1e-11, atol=1e-11)
    if "xm" in s:
        assert g["mp_iters"] == s["mp_iters"]
        assert abs(g["mp_step"] - s["mp_step"]) <= 1e-9 * s["mp_step"]
        np.testing.assert_allclose(g["xm"], s["xm"], rtol=tol, atol=tol)
        np.testing.assert_allclose(g["ym"], s["ym"], rtol=tol, atol=tol)


@pytest.mark.parametrize("overlap", ["0", "1"], ids=["reduce_scatter", "per_slice_reduce"])
@pytest.mark.parametrize("device_ids", [[0, 0], [0, 0, 0], [0] * 8], ids=["2", "3", "8"])
def test_shards_on_one_gpu_match_single_engine(gpu_required, monkeypatch, device_ids, overlap):
    # overlap=1: the exchange of A_p'y_p as P per-slice reductions on the comm streams
    # (these shards use the stream layout: the product is computed whole, then reduced slice by slice)
    monkeypatch.setenv("PDHG_DIST_OVERLAP", overlap)
    p = random_lp(30000, 20000, 6, seed=21)
    geng = HipPdhgEngine.from_problem(p, device_ids=device_ids)
    info = geng.dist_info()
    assert info["world"] == info["local_ranks"] == len(device_ids) and info["backend"] == 1
    g = _run(geng, p, 60, 25)
    s = _run(HipPdhgEngine.from_problem(p), p, 60, 25)
    _compare(g, s, p)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("shards", [2, 3])
def test_tiled_shards_match_single_engine_and_overlap_is_bitwise_neutral(gpu_required, monkeypatch, shards):
    """Large enough that each shard's A_p and A_p' use the tiled layout: A_p' is then cut at
    launched a residency round at a time and slice k is reduced to its owner as soon as its
    rows are complete, while the next round computes (the default for vectors this long).  With the peer back end both exchange patterns add the
    partials in rank order, so switching the overlap off must not change a single bit."""
    monkeypatch.setenv("PDHG_DIST_ROUND_WGS", "64")     # default granule: a residency round of 512 workgroups
    p = random_lp(1_100_000, 600_000, 5, seed=21)
    runs = {}
    # third run: the sweep's row groups dealt to the XCDs in contiguous eighths (what the builder does for banded matrices,
    # forced here), also inside the launches of one residency round each -- same row groups, same bits
    for overlap, remap in (("1", None), ("0", None), ("1", "1")):
        monkeypatch.setenv("PDHG_DIST_OVERLAP", overlap)
        if remap is None:
            monkeypatch.delenv("PDHG_TW_REMAP", raising=False)
        else:
            monkeypatch.setenv("PDHG_TW_REMAP", remap)
        geng = HipPdhgEngine.from_problem(p, device_ids=[0] * shards)
        info = geng.layout_info()
        # 2 shards: A_p' (600k x 600k) is tiled and launched in parts; 3 shards: its gathered
        # vector (400k rows of y) fits an XCD's L2, so it streams and is computed whole
        assert info["A_tiled_waves"] > 0 and (info["At_tiled_waves"] > 0) == (shards == 2)
        runs[overlap + (remap or "")] = _run(geng, p, 40, 10)
        geng.close()
    monkeypatch.delenv("PDHG_TW_REMAP", raising=False)
    for key, val in runs["1"].items():
        assert np.array_equal(np.asarray(val), np.asarray(runs["0"][key])), key
        assert np.array_equal(np.asarray(val), np.asarray(runs["11"][key])), key + " (XCD remap of the sweep)"
    s = _run(HipPdhgEngine.from_problem(p), p, 40, 10)
    _compare(runs["1"], s, p)


@pytest.mark.timeout(600)
def test_rccl_world_1_per_slice_reduce_on_tiled_layout(gpu_required, monkeypatch):
    """ncclReduce on the comm stream behind the cut A_p' product, 1-rank communicator."""
    monkeypatch.setenv("PDHG_DIST_OVERLAP", "1")
    monkeypatch.setenv("PDHG_DIST_ROUND_WGS", "64")
    p = random_lp(700_000, 600_000, 6, seed=9)
    geng = HipPdhgEngine.from_problem(p, device_ids=[0])
    assert geng.dist_info()["backend"] == 0 and geng.layout_info()["At_tiled_waves"] > 0
    g = _run(geng, p, 30, 5)
    s = _run(HipPdhgEngine.from_problem(p), p, 30, 5)
    _compare(g, s, p, tol=1e-10)


def test_row_partition_matches_the_host_rule(gpu_required):
    from firstorderlp_jl_amd.distributed import partition_rows, slice_stride
    p = H.skewed_lp(3000, 9000, seed=7, dense_rows=2, dense_cols=2)
    ranges = partition_rows(p.constraint_matrix, 3)
    eng = HipPdhgEngine.from_problem(p, device_ids=[0, 0, 0])
    info = eng.dist_info()
    assert (info["row_lo"], info["row_hi"]) == ranges[0]
    S = slice_stride(p.constraint_matrix.shape[1], 3)
    assert (info["col_lo"], info["col_hi"]) == (0, min(S, p.constraint_matrix.shape[1]))


@pytest.mark.parametrize("how", ["init_all", "init_rank", "init_rank_remote_route"])
def test_rccl_world_1_matches_single_engine(gpu_required, monkeypatch, how):
    """Real RCCL calls from the library: all-gather, reduce-scatter, broadcast on a 1-rank
    communicator.  "remote_route": the one-rank-per-process code paths (scalars of all ranks
    through ncclAllGather, vectors to the host through all-gather / grouped broadcasts) are
    taken although the only rank is local -- the routes `bench.py --gpus N` uses."""
    if how == "init_rank_remote_route":
        monkeypatch.setenv("PDHG_DIST_FORCE_REMOTE", "1")
    p = random_lp(20000, 15000, 8, seed=9)
    if how == "init_all":
        geng = HipPdhgEngine.from_problem(p, device_ids=[0])
    else:
        uid = HipPdhgEngine.dist_unique_id()
        assert len(uid) == 128
        geng = HipPdhgEngine.from_problem(p, device_id=0, unique_id=uid, rank=0, world=1)
    info = geng.dist_info()
    assert info["world"] == 1 and info["backend"] == 0
    g = _run(geng, p, 30, 10)
    s = _run(HipPdhgEngine.from_problem(p), p, 30, 10)
    _compare(g, s, p, tol=1e-10)
    y_new = np.abs(np.random.default_rng(0).standard_normal(geng.m))
    geng.set_current(None, y_new)
    seng = HipPdhgEngine.from_problem(p)
    seng.set_current(None, y_new)
    assert np.array_equal(geng.get_dual_product(), seng.get_dual_product())
    geng.close()


def _qp_problem():
    import scipy.sparse as sp
    p = random_lp(6000, 5000, 6, seed=31)
    B = sp.random(5000, 5000, density=0.001, random_state=6, format="csc")
    p.objective_matrix = sp.csc_matrix(B.T @ B + sp.diags(np.random.default_rng(8).uniform(0.0, 1.0, 5000)))
    return p


@pytest.mark.parametrize("device_ids", [[0, 0], [0]], ids=["p2p2", "rccl1"])
def test_qp_shards_match_single_engine(gpu_required, device_ids):
    p = _qp_problem()
    g = _run(HipPdhgEngine.from_problem(p, device_ids=device_ids), p, 40)
    s = _run(HipPdhgEngine.from_problem(p), p, 40)
    _compare(g, s, p)


# ---- the reference's KATs on groups ------------------------------------------------

def _group_factory(device_ids):
    def factory(problem):
        eng = HipPdhgEngine.from_problem(problem, device_ids=device_ids)
        assert eng.dist_info()["world"] == len(device_ids)
        return eng
    return factory


@pytest.mark.short_rows
@pytest.mark.parametrize("shards", [2, 3])
@pytest.mark.parametrize("case", kat_common.CASES, ids=lambda c: c.__name__)
def test_reference_kat_on_shard_group(gpu_required, case, shards):
    case(_group_factory([0] * shards))


@pytest.mark.short_rows
@pytest.mark.parametrize("case", kat_common.CASES, ids=lambda c: c.__name__)
def test_reference_kat_on_tiled_shard_group(gpu_required, monkeypatch, case):
    monkeypatch.setenv("PDHG_SPMV", "tiled")
    monkeypatch.setenv("PDHG_TILE_SHIFT", "8")
    case(_group_factory([0, 0]))


@pytest.mark.short_rows
@pytest.mark.parametrize("remote", ["0", "1"], ids=["local_route", "remote_route"])
@pytest.mark.parametrize("case", kat_common.CASES[:6], ids=lambda c: c.__name__)
def test_reference_kat_on_rccl_world_1(gpu_required, monkeypatch, case, remote):
    monkeypatch.setenv("PDHG_DIST_FORCE_REMOTE", remote)     # 1: the one-rank-per-process routes
    case(_group_factory([0]))


# ---- evaluation branch / rescaling on a group ---------------------------------------

def test_group_device_evaluation_matches_single_engine(gpu_required):
    from firstorderlp_jl_amd.evaluation import POINT_AVERAGE, POINT_CURRENT, POINT_RESTART
    p = H.skewed_lp(2500, 6000, 5)
    m, n = p.constraint_matrix.shape
    rng = np.random.default_rng(4)
    E, D = rng.uniform(0.5, 2.0, m), rng.uniform(0.5, 2.0, n)
    engines = [HipPdhgEngine.from_problem(p), HipPdhgEngine.from_problem(p, device_ids=[0, 0, 0]),
               HipPdhgEngine.from_problem(p, device_ids=[0])]
    os.environ["PDHG_DIST_FORCE_REMOTE"] = "1"       # RCCL world 1 through the one-rank-per-process routes
    try:
        engines.append(HipPdhgEngine.from_problem(p, device_ids=[0]))
    finally:
        del os.environ["PDHG_DIST_FORCE_REMOTE"]
    outs = []
    for eng in engines:
        eng.set_original_problem(E, D, p.objective_vector * D, p.right_hand_side * E,
                                 p.variable_lower_bound / D, p.variable_upper_bound / D)
        step, pw = H.initial_step_and_weight(p)
        st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
        for _ in range(12):
            take_step(ADAPTIVE, st)
        eng.save_restart_point()
        for _ in range(23):
            take_step(ADAPTIVE, st)
        rec = {}
        for point in (POINT_CURRENT, POINT_AVERAGE, POINT_RESTART):
            rec["eval", point] = eng.eval_point(point)
            rec["dist", point] = np.array(eng.distance_to_restart(point))
            rec["sumsq", point] = np.array(eng.point_sumsq(point))
            for rng_ in (0, 1, 2):
                rec["tr", point, rng_] = eng.trust_region_bound(point, 2.0, 0.5, 0.7, rng_)[:6]
            rec["tra", point] = eng.trust_region_bound(point, 2.0, 0.5, 0.7, 0, True)[:5]
            x, y = eng.get_point(point)
            rec["pt", point] = np.concatenate([x, y])
        rec["maxabs"] = np.array([eng.matrix_max_abs()])
        outs.append(rec)
    ref = outs[0]
    for other in outs[1:]:
        for key, val in ref.items():
            np.testing.assert_allclose(other[key], val, rtol=1e-9, atol=1e-9, err_msg=str(key))


@pytest.mark.parametrize("qp", [False, True], ids=["lp", "qp"])
def test_group_rescale_matches_single_engine(gpu_required, qp):
    p = _qp_problem() if qp else H.skewed_lp(3000, 5000, 11)
    single = HipPdhgEngine.from_problem(p)
    group = HipPdhgEngine.from_problem(p, device_ids=[0, 0, 0])
    rng = np.random.default_rng(1)
    x = rng.standard_normal(p.constraint_matrix.shape[1])
    y = rng.standard_normal(p.constraint_matrix.shape[0])
    # Ruiz: maxima are order-independent -> bit-identical factors, matrices and products
    es, ds = single.rescale(10, False, None)
    eg, dg = group.rescale(10, False, None)
    assert np.array_equal(es, eg) and np.array_equal(ds, dg)
    for a, b in zip(single.get_problem_vectors(), group.get_problem_vectors()):
        assert np.array_equal(a, b)
    assert np.array_equal(single.spmv(x), group.spmv(x))
    np.testing.assert_allclose(single.spmv_t(y), group.spmv_t(y), rtol=1e-12, atol=1e-12)
    assert single.matrix_max_abs() == group.matrix_max_abs()
    # L2 and Pock-Chambolle: column sums are added per shard, then over ranks
    es, ds = single.rescale(0, True, 1.0)
    eg, dg = group.rescale(0, True, 1.0)
    np.testing.assert_allclose(eg, es, rtol=1e-12)
    np.testing.assert_allclose(dg, ds, rtol=1e-12)
    np.testing.assert_allclose(single.spmv(x), group.spmv(x), rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("shards", [2, 3])
def test_group_with_column_chunk_passes_matches_the_plain_group(gpu_required, monkeypatch, shards):
    """Round 6 (csrc/dist.hpp: DistGroup::ag_chunks): with PDHG_DIST_AG_OVERLAP set, A_p xbar runs as one carried pass per
    column chunk against a chunk-major copy of xbar (on the peer back end: behind the ordinary all-gather).  A row's
    products are added chunk by chunk -- decisions of the plain group, iterates to 1e-9 -- and DEVICE RESCALING must reach
    the chunk layouts (their column factors travel in chunk layout): the same trajectory after Ruiz + Pock-Chambolle."""
    p = random_lp(30000, 20000, 6, seed=21)
    plain = HipPdhgEngine.from_problem(p, device_ids=[0] * shards)
    monkeypatch.setenv("PDHG_DIST_AG_OVERLAP", "1")
    chunked = HipPdhgEngine.from_problem(p, device_ids=[0] * shards)
    d = chunked.layout_describe()
    assert d["all_gather"]["chunks"] >= 2 and len(d["all_gather"]["passes"]) == d["all_gather"]["chunks"], d.get("all_gather")
    assert "all_gather" not in plain.layout_describe()
    for e in (plain, chunked):
        e.rescale(10, False, 1.0)
    g, s = _run(chunked, p, 40, 10), _run(plain, p, 40, 10)
    assert g["decisions"] == s["decisions"] and g["mp_iters"] == s["mp_iters"]
    for k in ("x", "y", "xa", "ya", "xm", "ym", "aty", "ax"):
        np.testing.assert_allclose(g[k], s[k], rtol=1e-9, atol=1e-9, err_msg=k)


@pytest.mark.timeout(900)
def test_group_solve_matches_single_engine_solve(gpu_required):
    """optimize() with solve_qp.jl's defaults, all on the device: three shards vs one handle."""
    from firstorderlp_jl_amd.distributed import multi_device_factory
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import optimize
    from tests.test_gpu_end_to_end import _params
    tol = 1e-6
    p = random_lp(12000, 10000, 8, seed=42)
    one = optimize(_params(tol, 40000), p)
    grp = optimize(_params(tol, 40000), p, multi_device_factory([0, 0, 0]))
    assert one.termination_string == grp.termination_string == "OPTIMAL"
    c1 = one.iteration_stats[-1].convergence_information[0]
    c3 = grp.iteration_stats[-1].convergence_information[0]
    scale = 1.0 + abs(c1.primal_objective)
    assert abs(c3.primal_objective - c1.primal_objective) <= 50 * tol * scale
    assert abs(c3.dual_objective - c1.dual_objective) <= 50 * tol * scale
    assert 0.5 <= grp.iteration_count / one.iteration_count <= 2.0


# ---- round 3: one issuing host thread per shard, rank-local ingest, RCCL binding --------------

@pytest.mark.parametrize("overlap", ["0", "1"], ids=["reduce_scatter", "per_slice_reduce"])
@pytest.mark.parametrize("device_ids", [[0, 0], [0] * 8], ids=["2", "8"])
def test_thread_per_shard_issue_is_bitwise_the_single_thread_issue(gpu_required, monkeypatch, device_ids, overlap):
    """pdhg_create_multi issues every shard's trial from its own host thread (ShardPool);
    PDHG_SHARD_THREADS=0 issues all of them from the caller.  Same launches, same order per
    stream, rank-ordered sums: not a bit may differ.  (Peer-kernel back end here; the RCCL
    back end takes the same route on >= 2 GPUs, tests/test_gpu_multi_device.py.)"""
    monkeypatch.setenv("PDHG_DIST_OVERLAP", overlap)
    p = random_lp(30000, 20000, 6, seed=21)
    runs = {}
    for threads in ("1", "0"):
        monkeypatch.setenv("PDHG_SHARD_THREADS", threads)
        eng = HipPdhgEngine.from_problem(p, device_ids=device_ids)
        runs[threads] = _run(eng, p, 60, 25)
        trials, issue, wait = eng.host_issue_stats()
        assert trials >= 60 and issue > 0.0
        eng.close()
    for key, val in runs["1"].items():
        assert np.array_equal(np.asarray(val), np.asarray(runs["0"][key])), key


def test_thread_per_shard_qp_group(gpu_required):
    kat_common.quadratic_programming_1(_group_factory([0, 0, 0]))
    kat_common.malitsky_pock_smoothing(_group_factory([0, 0, 0]))


@pytest.mark.parametrize("ingest", ["global", "rows"])
def test_one_process_per_gpu_route_with_one_rank(gpu_required, ingest):
    """tests/workers/dist_rank_worker.py under torch.distributed.run with ONE rank: gloo group,
    id broadcast, pdhg_create_dist (global ingest) / pdhg_create_dist_rows (rank-local
    ingest), 1-rank RCCL communicator, compared with the single handle inside the worker."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", "29611",
           os.path.join(root, "tests", "workers", "dist_rank_worker.py"), ingest, "0"]
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0 and "dist worker ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_rank_local_ingest_equals_global_ingest_bitwise(gpu_required):
    """pdhg_create_dist_rows on the rank's own rows == pdhg_create_dist on the global matrix
    (1-rank communicator: the slice is the whole matrix, the code path is the rank-local one)."""
    from firstorderlp_jl_amd.distributed import row_shard_of
    p = random_lp(9000, 7000, 7, seed=3)
    uid = HipPdhgEngine.dist_unique_id()
    a = HipPdhgEngine.from_problem(p, unique_id=uid, rank=0, world=1, device_id=0)
    ra = _run(a, p, 40, 10)
    a.close()
    sh = row_shard_of(p, HipPdhgEngine.partition_rows(p.constraint_matrix, 1), 0)
    b = HipPdhgEngine.from_row_shard(sh["m_global"], sh["row_bounds"], sh["constraint_rows"], sh["objective_vector"],
                                     sh["right_hand_side_rows"], sh["variable_lower_bound"], sh["variable_upper_bound"],
                                     sh["num_equalities"], HipPdhgEngine.dist_unique_id(), 0, 1, device_id=0)
    rb = _run(b, p, 40, 10)
    for key, val in ra.items():
        assert np.array_equal(np.asarray(val), np.asarray(rb[key])), key


def test_rccl_binding_is_reported_and_version_checked(gpu_required):
    info = HipPdhgEngine.rccl_info()
    assert info["path"].endswith(".so") or ".so." in info["path"], info
    assert info["runtime_version"] // 10000 == info["compiled_version"] // 10000 == 2, info
    print("RCCL bound at run time:", info)


def test_batched_take_steps_on_a_shard_group(gpu_required):
    """pdhg_take_steps_adaptive on a 2-shard group (peer back end on one GPU): n steps in one call == n single calls."""
    from firstorderlp_jl_amd.primal_dual_hybrid_gradient import (AdaptiveStepsizeParams, PdhgSolverState, take_step,
                                                                 take_steps)
    p = random_lp(3000, 2500, 6, seed=13)
    policy = AdaptiveStepsizeParams(0.3, 0.6)
    step, pw = H.initial_step_and_weight(p)
    outs = []
    for batched in (False, True):
        eng = HipPdhgEngine.from_problem(p, device_ids=[0, 0])
        st = PdhgSolverState(eng, step_size=step, primal_weight=pw)
        if batched:
            assert take_steps(policy, st, 40) == 40
        else:
            for _ in range(40):
                take_step(policy, st)
        outs.append((np.concatenate(eng.get_current()), np.concatenate(eng.get_average()), st.step_size,
                     st.total_number_iterations, st.cumulative_kkt_passes))
        eng.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2:] == outs[1][2:]


# ---- round 4: a group's trial as one persistent kernel per shard with cross-shard barriers (group_kernel.hpp) ---------

@pytest.mark.parametrize("maker", [lambda: random_lp(30000, 20000, 6, seed=21),
                                   lambda: H.skewed_lp(9000, 12000, seed=5, dense_rows=2, dense_cols=2, base_nnz=6)],
                         ids=["random", "long_rows"])
@pytest.mark.parametrize("device_ids", [[0, 0], [0, 0, 0], [0] * 4], ids=["2", "3", "4"])
def test_group_trial_kernels_are_bitwise_the_per_launch_group_path(gpu_required, monkeypatch, device_ids, maker):
    """Shards on one device: the trial as ONE persistent kernel per shard -- xbar stored into every shard's buffer, the
    partial A_p'y_p added by the slice's owner in rank order, two cross-shard barriers inside the kernels -- against the
    ordinary group path (launch by launch, exchange kernels between cross-stream events).  Same element arithmetic, the
    same rank order in the reduce, exactly rounded acceptance sums: not a bit may differ -- adaptive steps, the
    Malitsky-Pock split (xbar-only retries), the lazy average update, restart to the average."""
    p = maker()
    runs = {}
    for coop in ("1", "0"):
        monkeypatch.setenv("PDHG_GROUP_COOP", coop)
        eng = HipPdhgEngine.from_problem(p, device_ids=device_ids)
        runs[coop] = _run(eng, p, 60, 25)
        runs[coop]["launches"] = eng.layout_info()["group_coop_trials"]
        eng.close()
    assert runs["1"].pop("launches") >= 60 and runs["0"].pop("launches") == 0          # the new path really ran
    for key, val in runs["1"].items():
        assert np.array_equal(np.asarray(val), np.asarray(runs["0"][key])), key
    s = _run(HipPdhgEngine.from_problem(p), p, 60, 25)
    _compare(runs["1"], s, p)


def test_group_trial_kernels_fall_back_when_a_barrier_cannot_complete(gpu_required, monkeypatch):
    """More workgroups than the device holds side by side (test knob): the first cross-shard barrier times out, the
    kernels raise their error words, the host repeats the trial launch by launch and keeps the group there -- same bits."""
    p = random_lp(30000, 20000, 6, seed=22)
    monkeypatch.setenv("PDHG_GROUP_COOP", "0")
    ref = _run(HipPdhgEngine.from_problem(p, device_ids=[0, 0]), p, 12)
    monkeypatch.setenv("PDHG_GROUP_COOP", "1")
    monkeypatch.setenv("PDHG_COOP_TEST_PRETEND_WGS", "4096")
    eng = HipPdhgEngine.from_problem(p, device_ids=[0, 0])
    got = _run(eng, p, 12)
    assert eng.layout_info()["group_coop_fallbacks"] == 1
    for key, val in got.items():
        assert np.array_equal(np.asarray(val), np.asarray(ref[key])), key
