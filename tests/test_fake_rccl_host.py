"""The test-only RCCL stand-in (tests/fake_rccl) checked ON ITS OWN, on the CPU: three processes, host buffers
(FAKE_RCCL_HOST_BUFFERS=1: no HIP call is made), vectors longer than a staging slot, in-place forms exactly as
csrc/dist.hpp issues them (all-gather of a slice inside the full buffer, reduce-scatter into the own slice, reduce to a
root in place, broadcast in place, a group of broadcasts), rank-ordered sums.  The GPU legs that use it
(tests/test_gpu_fake_rccl.py) then only have the product library left to suspect."""
import ctypes
import multiprocessing as mp
import os

import numpy as np
import pytest

from tests import fake_rccl

NCCL_DOUBLE, NCCL_SUM, NCCL_MAX = 8, 0, 2       # rccl.h: ncclFloat64 = 8, ncclSum = 0, ncclMax = 2


class UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


def _load():
    L = ctypes.CDLL(fake_rccl.build())
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    L.ncclCommInitRank.argtypes = [ctypes.POINTER(vp), ci, UniqueId, ci]
    L.ncclCommDestroy.argtypes = [vp]
    L.ncclAllGather.argtypes = [vp, vp, sz, ci, vp, vp]
    L.ncclReduceScatter.argtypes = [vp, vp, sz, ci, ci, vp, vp]
    L.ncclReduce.argtypes = [vp, vp, sz, ci, ci, ci, vp, vp]
    L.ncclBroadcast.argtypes = [vp, vp, sz, ci, ci, vp, vp]
    L.ncclGetErrorString.restype = ctypes.c_char_p
    return L


def _data(rank, n):
    rng = np.random.default_rng(100 + rank)
    return rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8, n)


def _rank_main(rank, world, idbytes, n, q):
    try:
        os.environ["FAKE_RCCL_HOST_BUFFERS"] = "1"
        os.environ["FAKE_RCCL_SLOT_MB"] = "1"
        os.environ["FAKE_RCCL_TIMEOUT_S"] = "60"
        L = _load()
        uid = UniqueId()
        ctypes.memmove(ctypes.byref(uid), idbytes, 128)
        comm = ctypes.c_void_p()
        assert L.ncclCommInitRank(ctypes.byref(comm), world, uid, rank) == 0
        S = (n + world - 1) // world
        ptr = lambda a, off=0: ctypes.c_void_p(a.ctypes.data + 8 * off)      # noqa: E731
        every = [_data(r, world * S) for r in range(world)]
        # all-gather in place: my slice of the full buffer
        buf = np.zeros(world * S)
        buf[rank * S:(rank + 1) * S] = every[rank][rank * S:(rank + 1) * S]
        assert L.ncclAllGather(ptr(buf, rank * S), ptr(buf), S, NCCL_DOUBLE, comm, None) == 0
        want = np.concatenate([every[r][r * S:(r + 1) * S] for r in range(world)])
        assert np.array_equal(buf, want), "all-gather"
        # reduce-scatter in place: rank-ordered sum of everybody's slice `rank`
        for op in (NCCL_SUM, NCCL_MAX):
            buf = every[rank].copy()
            assert L.ncclReduceScatter(ptr(buf), ptr(buf, rank * S), S, NCCL_DOUBLE, op, comm, None) == 0
            acc = every[0][rank * S:(rank + 1) * S].copy()
            for r in range(1, world):
                acc = acc + every[r][rank * S:(rank + 1) * S] if op == NCCL_SUM else np.fmax(acc, every[r][rank * S:(rank + 1) * S])
            assert np.array_equal(buf[rank * S:(rank + 1) * S], acc), "reduce-scatter"
            others = np.ones(world * S, bool)
            others[rank * S:(rank + 1) * S] = False
            assert np.array_equal(buf[others], every[rank][others]), "reduce-scatter touched foreign slices"
        # per-slice reduce to its owner, in place (csrc/dist.hpp: dist_reduce_slice_async)
        buf = every[rank].copy()
        for k in range(world):
            assert L.ncclReduce(ptr(buf, k * S), ptr(buf, k * S), S, NCCL_DOUBLE, NCCL_SUM, k, comm, None) == 0
        acc = every[0][rank * S:(rank + 1) * S].copy()
        for r in range(1, world):
            acc = acc + every[r][rank * S:(rank + 1) * S]
        assert np.array_equal(buf[rank * S:(rank + 1) * S], acc), "reduce"
        # a group of in-place broadcasts of ragged row ranges (dist_all_gather_rows), one of them empty
        bounds = [0, n // 5, n // 5, n]
        bounds = bounds[:world] + [n] if world < 3 else bounds + [n] * (world - 3)
        buf = np.full(n, -1.0)
        buf[bounds[rank]:bounds[rank + 1]] = every[rank][bounds[rank]:bounds[rank + 1]]
        assert L.ncclGroupStart() == 0
        for r in range(world):
            cnt = bounds[r + 1] - bounds[r]
            if cnt > 0:
                assert L.ncclBroadcast(ptr(buf, bounds[r]), ptr(buf, bounds[r]), cnt, NCCL_DOUBLE, r, comm, None) == 0
        assert L.ncclGroupEnd() == 0
        want = np.concatenate([every[r][bounds[r]:bounds[r + 1]] for r in range(world)])
        assert np.array_equal(buf, want), "broadcast"
        # the 32 scalars of a trial (combine_scalars): a short all-gather out of place
        mine = _data(rank + 50, 32)
        allv = np.zeros(32 * world)
        assert L.ncclAllGather(ptr(mine), ptr(allv), 32, NCCL_DOUBLE, comm, None) == 0
        assert np.array_equal(allv, np.concatenate([_data(r + 50, 32) for r in range(world)]))
        stats = (ctypes.c_longlong * 2)()
        L.fake_rccl_stats(comm, stats)
        assert L.ncclCommDestroy(comm) == 0
        q.put((rank, "ok", int(stats[0])))
    except BaseException as exc:      # noqa: BLE001 -- reported to the parent
        q.put((rank, repr(exc), 0))


@pytest.mark.parametrize("world", [1, 2, 3])
def test_collectives_between_processes_on_host_buffers(world):
    L = _load()
    uid = UniqueId()
    assert L.ncclGetUniqueId(ctypes.byref(uid)) == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n = 300_001            # 2.4 MB: three chunks of a 1 MiB slot, the last one ragged
    procs = [ctx.Process(target=_rank_main, args=(r, world, bytes(uid), n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), res
    assert not [f for f in os.listdir("/tmp") if f.startswith(uid.internal.decode())], "rendezvous file left behind"


def test_a_missing_rank_is_an_error_not_a_hang(monkeypatch):
    monkeypatch.setenv("FAKE_RCCL_TIMEOUT_S", "1")
    monkeypatch.setenv("FAKE_RCCL_HOST_BUFFERS", "1")
    L = _load()
    uid = UniqueId()
    assert L.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rc = L.ncclCommInitRank(ctypes.byref(comm), 2, uid, 0)
    assert rc != 0 and b"fake_rccl" in L.ncclGetErrorString(rc)
    assert L.ncclCommInitAll(None, 2, None) != 0
