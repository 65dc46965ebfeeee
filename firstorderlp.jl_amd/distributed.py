"""1-D row partition of the constraint matrix across the GPUs of one node.

PRODUCT PATH: the exchange lives inside the HIP library (csrc/dist.hpp: RCCL
reduce-scatter / all-gather over xGMI, or direct peer kernels inside one
process).  ``make_row_partitioned_hip_engine`` only hands the library a
communicator id; the engine it returns is an ordinary ``HipPdhgEngine`` whose
every method keeps global vector lengths.  No torch.distributed call is on the
iteration path.

    one process per GPU :  HipPdhgEngine(..., unique_id=, rank=, world=)   pdhg_create_dist
    one process, N GPUs :  HipPdhgEngine(..., device_ids=[0, 1, ...])      pdhg_create_multi

HOST MIRROR (``RowPartitionedEngine``): the same algorithm -- row shards,
owned column slices, rank-ordered sums, scalars combined in rank order --
restated in numpy over a local engine object and a small collective interface,
so that the world_size > 1 logic can be exercised on a box without GPUs
(tests/test_distributed_gloo.py: gloo, the CPU oracle injected as the local
engine).  It mirrors csrc/dist.hpp step by step and is not used on GPUs.

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


def row_partitioned_factory(group=None):
    """The same for one process per GPU (torch.distributed group for the id hand-out)."""
    def factory(problem):
        return make_row_partitioned_hip_engine(problem, group=group)
    factory.takes_original_problem = True
    return factory


# ---- host mirror (CPU tests) ----------------------------------------------------------

class TorchComm:
    """The collectives the mirror needs, on host numpy arrays over
    torch.distributed (gloo in the CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)

    def all_gather(self, arr):
        """[world_size, len(arr)] float64: every rank's ``arr`` (equal lengths)."""
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64))
        outs = [torch.empty_like(t) for _ in range(self.world_size)]
        self.dist.all_gather(outs, t, group=self.group)
        return np.stack([o.numpy() for o in outs])

    def all_gather_rows(self, arr, sizes):
        """Concatenate per-rank slices of different lengths (row vectors)."""
        width = max(max(sizes), 1)
        buf = np.zeros(width)
        buf[:len(arr)] = arr
        parts = self.all_gather(buf)
        return np.concatenate([parts[r, :s] for r, s in enumerate(sizes)])


class RowPartitionedEngine:
    """numpy mirror of csrc/dist.hpp over a local engine that offers the shard
    primitives (tests/oracle_engine.OracleEngine): ``dist_trial_begin`` /
    ``dist_trial_dual_begin`` (x', xbar, y'_p and the partial t_p = A_p' y'_p),
    ``exchange_array()`` (t_p, n doubles, in place), ``dist_trial_end_slice``."""

    def __init__(self, local, comm, row_ranges):
        self.local = local
        self.comm = comm
        self.row_ranges = list(row_ranges)
        self.lo, self.hi = self.row_ranges[comm.rank]
        self.sizes = [hi - lo for lo, hi in self.row_ranges]
        self.n = local.n
        self.m = self.row_ranges[-1][1]
        assert local.m == self.hi - self.lo
        self.S = slice_stride(self.n, comm.world_size)
        self.clo = min(self.n, comm.rank * self.S)
        self.chi = min(self.n, (comm.rank + 1) * self.S)

    # ---- the exchange: reduce-scatter (rank-order sum on the owned slice), then
    # all-gather of the slices (the local engine keeps full-length vectors)
    def _reduce_scatter_all_gather(self, partial):
        world, S, n = self.comm.world_size, self.S, self.n
        padded = np.zeros(world * S)
        padded[:n] = partial
        parts = self.comm.all_gather(padded)                       # [world, world*S]
        own = parts[0, self.comm.rank * S:(self.comm.rank + 1) * S].copy()
        for r in range(1, world):                                  # ranks ascending: p2p_reduce_kernel
            own = own + parts[r, self.comm.rank * S:(self.comm.rank + 1) * S]
        slices = self.comm.all_gather(own)                         # [world, S]
        partial[:] = slices.reshape(-1)[:n]

    def _combine(self, raw_local):
        """Scalars of all ranks added in rank order on every rank (combine_scalars);
        the replicated QP term [4] is taken once."""
        raws = self.comm.all_gather(np.asarray(raw_local, dtype=np.float64))
        out = raws[0].copy()
        for r in range(1, self.comm.world_size):
            out[:4] = out[:4] + raws[r, :4]
            out[4] = max(out[4], raws[r, 4])
        return out

    def _finish_trial(self):
        self._reduce_scatter_all_gather(self.local.exchange_array())
        return self._combine(self.local.dist_trial_end_slice(self.clo, self.chi))

    # ---- hot path ----
    def trial_step(self, step_size, primal_weight, theta=1.0):
        self.local.dist_trial_begin(step_size, primal_weight, theta)
        return self._finish_trial()

    def accept(self, avg_weight):
        self.local.accept(avg_weight)

    # Malitsky-Pock (pdhg.jl:555-647): the primal half is slice-local, every
    # linesearch iteration costs one exchange like an adaptive trial.
    def trial_primal(self, step_size, primal_weight):
        self.local.trial_primal(step_size, primal_weight)

    def trial_dual(self, step_size, primal_weight, theta):
        self.local.dist_trial_dual_begin(step_size, primal_weight, theta)
        return self._finish_trial()

    def add_current_primal_to_average(self, weight):
        self.local.add_current_primal_to_average(weight)

    def _refresh_dual_product(self):
        self.local.dist_dual_product_begin()
        self._reduce_scatter_all_gather(self.local.exchange_array())
        self.local.dist_dual_product_end()

    # ---- average / restart ----
    def average_info(self):
        return self.local.average_info()

    def get_average(self):
        xa, ya = self.local.get_average()
        return xa, self.comm.all_gather_rows(ya, self.sizes)

    def reset_average(self):
        self.local.reset_average()

    def restart_to_average(self):
        self.local.restart_to_average()
        self._refresh_dual_product()

    # ---- iterate I/O ----
    def get_current(self):
        x, y = self.local.get_current()
        return x, self.comm.all_gather_rows(y, self.sizes)

    def get_dual_product(self):
        return self.local.get_dual_product()

    def set_current(self, x=None, y=None):
        self.local.set_current(x, None if y is None else y[self.lo:self.hi])
        self._refresh_dual_product()

    # ---- standalone mat-vecs for the host evaluation branch ----
    def spmv(self, x):
        """A*x: every rank multiplies its row block, slices are concatenated."""
        return self.comm.all_gather_rows(self.local.spmv(x), self.sizes)

    def spmv_t(self, y):
        """A'*y = sum_p A_p' y_p: local partial, then the rank-ordered sum."""
        partial = np.array(self.local.spmv_t(y[self.lo:self.hi]), dtype=np.float64)
        self._reduce_scatter_all_gather(partial)
        return partial

    def close(self):
        self.local.close()
