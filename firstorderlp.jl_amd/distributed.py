"""1-D row partition of the constraint matrix across the GPUs of one node.

PRODUCT PATH: the exchange lives inside the HIP library (csrc/dist.hpp: RCCL
reduce-scatter / all-gather over xGMI, or direct peer kernels inside one
process).  ``make_row_partitioned_hip_engine`` only hands the library a
communicator id; the engine it returns is an ordinary ``HipPdhgEngine`` whose
every method keeps global vector lengths.  No torch.distributed call is on the
iteration path.

    one process per GPU :  HipPdhgEngine(..., unique_id=, rank=, world=)   pdhg_create_dist
    one process, N GPUs :  HipPdhgEngine(..., device_ids=[0, 1, ...])      pdhg_create_multi

The numpy restatement of this exchange that the world_size-2 CPU tests drive over gloo (row shards, owned
column slices, rank-ordered sums) is test infrastructure and lives in tests/dist_mirror.py.

The reference has no counterpart (single process, single thread); the
arithmetic being distributed is src/primal_dual_hybrid_gradient.jl:442-549.
"""
import os

import numpy as np

from .quadratic_programming import as_csc


def partition_rows(constraint_matrix, world_size):
    """Contiguous row ranges [(lo, hi)] * world_size balanced by nonzeros
    (the same rule as partition_rows_by_nnz in csrc/dist.hpp)."""
    csr_indptr = np.zeros(constraint_matrix.shape[0] + 1, dtype=np.int64)
    np.add.at(csr_indptr, constraint_matrix.indices + 1, 1)
    np.cumsum(csr_indptr, out=csr_indptr)
    m = constraint_matrix.shape[0]
    nnz = int(csr_indptr[-1])
    bounds = [0]
    for p in range(1, world_size):
        # first r with prefix[r] * world >= nnz * p  (exact integer compare)
        r = int(np.searchsorted(csr_indptr * world_size, nnz * p, side="left"))
        r = min(max(r, bounds[-1]), m)
        bounds.append(r)
    bounds.append(m)
    return [(bounds[p], bounds[p + 1]) for p in range(world_size)]


def slice_stride(n, world_size):
    """Column-slice stride S: rank r owns columns [r*S, min(n, (r+1)*S))
    (init_group_geometry in csrc/pdhg_hip.hip)."""
    per = (n + world_size - 1) // world_size
    return max(16, (per + 15) // 16 * 16)


def shard_rows(problem, lo, hi):
    """The arguments of a local engine for rows [lo, hi) of ``problem``."""
    A = problem.constraint_matrix.tocsr()[lo:hi, :]
    Q = getattr(problem, "objective_matrix", None)
    return dict(
        objective_matrix=Q if (Q is not None and Q.nnz > 0) else None,   # replicated on every rank
        constraint_matrix=as_csc(A),
        objective_vector=problem.objective_vector,
        right_hand_side=problem.right_hand_side[lo:hi],
        variable_lower_bound=problem.variable_lower_bound,
        variable_upper_bound=problem.variable_upper_bound,
        num_equalities=int(min(max(problem.num_equalities - lo, 0), hi - lo)),
    )


# ---- product path -----------------------------------------------------------------

def broadcast_unique_id(rank, group=None):
    """Rank 0 creates the RCCL id inside the library; the 128 bytes travel over
    whatever process group the host already has (gloo is enough)."""
    import torch
    import torch.distributed as dist
    from .engine import HipPdhgEngine
    from . import _lib
    buf = torch.zeros(_lib.UNIQUE_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        buf = torch.frombuffer(bytearray(HipPdhgEngine.dist_unique_id()), dtype=torch.uint8).clone()
    backend = dist.get_backend(group)
    if backend == "nccl":
        buf = buf.cuda()
    dist.broadcast(buf, src=0, group=group)
    return bytes(buf.cpu().numpy().tobytes())


def make_row_partitioned_hip_engine(problem, device_id=None, group=None):
    """One process per GPU: this rank's shard of ``problem`` behind an ordinary
    HipPdhgEngine (pdhg_create_dist).  Needs an initialised torch.distributed
    group only to hand out the communicator id."""
    import torch.distributed as dist
    from .engine import HipPdhgEngine
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if device_id is None:
        device_id = int(os.environ.get("LOCAL_RANK", rank))
    uid = broadcast_unique_id(rank, group)
    return HipPdhgEngine.from_problem(problem, device_id=device_id, unique_id=uid,
                                      rank=rank, world=world)


def make_row_shard_hip_engine(shard, device_id=None, group=None):
    """One process per GPU with rank-local ingest (pdhg_create_dist_rows): ``shard`` is what
    ``row_shard_of`` produced for THIS rank -- only its rows of the matrix and of b plus the
    global n-vectors -- so no rank ever holds the whole matrix."""
    import torch.distributed as dist
    from .engine import HipPdhgEngine
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if device_id is None:
        device_id = int(os.environ.get("LOCAL_RANK", rank))
    uid = broadcast_unique_id(rank, group)
    return HipPdhgEngine.from_row_shard(
        shard["m_global"], shard["row_bounds"], shard["constraint_rows"], shard["objective_vector"],
        shard["right_hand_side_rows"], shard["variable_lower_bound"], shard["variable_upper_bound"],
        shard["num_equalities"], uid, rank, world, device_id=device_id,
        objective_matrix=shard.get("objective_matrix"))


def row_shard_of(problem, row_bounds, rank):
    """Rank ``rank``'s slice of ``problem`` for ``make_row_shard_hip_engine`` (rows
    row_bounds[rank]..row_bounds[rank+1] in CSR order -- slicing is a view-cheap operation
    there -- and the global vectors)."""
    lo, hi = int(row_bounds[rank]), int(row_bounds[rank + 1])
    A = problem.constraint_matrix
    rows = (A if A.format == "csr" else A.tocsr())[lo:hi, :]
    Q = getattr(problem, "objective_matrix", None)
    return dict(m_global=int(A.shape[0]), row_bounds=np.asarray(row_bounds, dtype=np.int64),
                constraint_rows=rows, objective_vector=problem.objective_vector,
                right_hand_side_rows=np.ascontiguousarray(problem.right_hand_side[lo:hi]),
                variable_lower_bound=problem.variable_lower_bound,
                variable_upper_bound=problem.variable_upper_bound,
                num_equalities=int(problem.num_equalities),
                objective_matrix=Q if (Q is not None and Q.nnz > 0) else None)


def make_multi_device_hip_engine(problem, device_ids):
    """One process driving several GPUs (pdhg_create_multi)."""
    from .engine import HipPdhgEngine
    return HipPdhgEngine.from_problem(problem, device_ids=list(device_ids))


def multi_device_factory(device_ids):
    """``optimize(params, problem, multi_device_factory([0, 1, ...]))``: the engine is
    built from the ORIGINAL problem and rescaled on the devices."""
    def factory(problem):
        return make_multi_device_hip_engine(problem, device_ids)
    factory.takes_original_problem = True
    return factory


def row_partitioned_factory(group=None, device_id=None):
    """The same for one process per GPU (torch.distributed group for the id hand-out).
    ``device_id``: this rank's GPU (default: LOCAL_RANK)."""
    def factory(problem):
        return make_row_partitioned_hip_engine(problem, device_id=device_id, group=group)
    factory.takes_original_problem = True
    return factory
