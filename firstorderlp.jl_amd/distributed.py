"""1-D row partition of the constraint matrix across the GPUs of one node:
one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI)
for the single exchange step per trial.

Rank p holds the contiguous row block A_p (balanced by nnz, equalities-first
order preserved), its slices of y / b / sum_y, and a full replica of the
n-vectors.  Per trial step (see include/pdhg_hip.h, "row-partitioned form"):

    begin : x', xbar (replicated, identical on every rank), y'_p, and the
            local partial A_p' y'_p  -> exchange buffer [0..n), sum dy_p^2 -> [n]
    all_reduce(sum) of the n+1 doubles                      <- the only collective
    end   : dx.(A'y'-A'y), |dx|^2, |A'y'-A'y|^2 on the replicated vectors

Every rank ends with bitwise-identical A'y' (the all-reduce delivers one
result to all ranks) and therefore takes identical accept/reject decisions.
The reference has no counterpart (single process, single thread).
"""
import os

import numpy as np

from .quadratic_programming import as_csc


def partition_rows(constraint_matrix, world_size):
    """Contiguous row ranges [(lo, hi)] * world_size balanced by nonzeros."""
    csr_indptr = np.zeros(constraint_matrix.shape[0] + 1, dtype=np.int64)
    np.add.at(csr_indptr, constraint_matrix.indices + 1, 1)
    np.cumsum(csr_indptr, out=csr_indptr)
    m = constraint_matrix.shape[0]
    nnz = int(csr_indptr[-1])
    bounds = [0]
    for p in range(1, world_size):
        target = nnz * p / world_size
        r = int(np.searchsorted(csr_indptr, target, side="left"))
        r = min(max(r, bounds[-1]), m)
        bounds.append(r)
    bounds.append(m)
    return [(bounds[p], bounds[p + 1]) for p in range(world_size)]


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


class TorchComm:
    """torch.distributed collectives (RCCL on GPU, gloo in the CPU tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)

    def all_reduce_sum(self, tensor):
        self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM, group=self.group)

    def all_reduce_sum_async(self, tensor):
        """Start the all-reduce and return its work handle: with RCCL the collective
        is ordered after what the current stream has queued so far and runs on the
        communicator's stream, so kernels launched next overlap it; ``wait()`` orders
        the current stream after it."""
        return self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM, group=self.group,
                                    async_op=True)

    def all_reduce_min_host(self, arr):
        """Element-wise minimum of a small int64 numpy vector over ranks (set-up only)."""
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64)).to(dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return t.cpu().numpy()

    def all_reduce_host(self, arr):
        """Sum a float64 numpy vector over ranks (evaluation cadence only)."""
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).to(dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def all_gather_host(self, arr, sizes):
        """Concatenate per-rank float64 numpy slices (evaluation cadence only)."""
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        width = max(max(sizes), 1)
        buf = torch.zeros(width, dtype=torch.float64, device=dev)
        buf[:len(arr)] = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
        outs = [torch.empty_like(buf) for _ in range(self.world_size)]
        self.dist.all_gather(outs, buf, group=self.group)
        return np.concatenate([o[:s].cpu().numpy() for o, s in zip(outs, sizes)])


class _DeviceBuffer:
    """Zero-copy view of library-owned device memory for torch.as_tensor."""

    def __init__(self, ptr, length):
        self.__cuda_array_interface__ = {
            "shape": (int(length),), "typestr": "<f8", "data": (int(ptr), False),
            "version": 2, "strides": None}


def hip_exchange_tensor(engine, cache):
    """torch view of the HIP engine's current exchange buffer (n+1 doubles)."""
    import torch
    ptr = engine.dist_exchange_ptr()
    t = cache.get(ptr)
    if t is None:
        t = torch.as_tensor(_DeviceBuffer(ptr, engine.n + 1), device="cuda")
        assert t.data_ptr() == ptr, "torch copied the exchange buffer"
        cache[ptr] = t
    return t


class RowPartitionedEngine:
    """Presents the ``HipPdhgEngine`` interface the host driver uses, over a
    local row-shard engine plus the all-reduce.  ``local`` must provide the
    dist_* methods and ``exchange_tensor()``."""

    def __init__(self, local, comm, row_ranges):
        self.local = local
        self.comm = comm
        self.row_ranges = list(row_ranges)
        self.lo, self.hi = self.row_ranges[comm.rank]
        self.sizes = [hi - lo for lo, hi in self.row_ranges]
        self.n = local.n
        self.m = self.row_ranges[-1][1]
        assert local.m == self.hi - self.lo
        self._bounds = None

    # ---- hot path ----
    def _parts(self):
        """Column ranges of the exchange buffer, agreed by all ranks once.  Every
        rank cuts its own A_p' by ITS workgroups (pdhg_dist_parts), so the local
        limits differ from rank to rank; a range may be exchanged once every rank
        has finished it, hence the element-wise minimum of the limits -- and no
        cutting at all unless every rank can cut into the same number of parts."""
        if self._bounds is None:
            want = int(os.environ.get("PDHG_DIST_PARTS", "4"))
            useful = self.comm.world_size > 1 or "PDHG_DIST_PARTS" in os.environ   # nothing to overlap alone
            local = list(self.local.dist_parts(want)) if (want > 1 and useful) else [0, self.n]
            k = len(local) - 1
            kmin, neg_kmax = self.comm.all_reduce_min_host(np.array([k, -k]))
            if kmin != -neg_kmax or kmin < 2:
                if k > 1:
                    self.local.dist_parts(1)         # the library launches what it handed out last
                self._bounds = [0, self.n]
            else:
                agreed = self.comm.all_reduce_min_host(np.array(local))
                self._bounds = [0] + [int(b) for b in agreed[1:-1]] + [self.n]
        return self._bounds

    def _exchange_in_parts(self, begin_part, step_size, primal_weight, theta):
        """Part k's columns are all-reduced while part k+1 is still being computed
        (A_p'y'_p is produced range by range); one collective per part, the last
        one also carries slot [n]."""
        bounds = self._parts()
        nparts = len(bounds) - 1
        works = []
        for k in range(nparts):
            begin_part(step_size, primal_weight, theta, k, nparts)
            t = self.local.exchange_tensor()
            hi = bounds[k + 1] + (1 if k == nparts - 1 else 0)
            works.append(self.comm.all_reduce_sum_async(t[bounds[k]:hi]))
        for w in works:
            w.wait()

    def trial_step(self, step_size, primal_weight, theta=1.0):
        if len(self._parts()) > 2:
            self._exchange_in_parts(self.local.dist_trial_begin_part, step_size, primal_weight, theta)
        else:
            self.local.dist_trial_begin(step_size, primal_weight, theta)
            self.comm.all_reduce_sum(self.local.exchange_tensor())
        return self.local.dist_trial_end()

    def accept(self, avg_weight):
        self.local.accept(avg_weight)

    # Malitsky-Pock (pdhg.jl:555-647): the primal half is rank-local, every
    # linesearch iteration costs one all-reduce like an adaptive trial.
    def trial_primal(self, step_size, primal_weight):
        self.local.trial_primal(step_size, primal_weight)

    def trial_dual(self, step_size, primal_weight, theta):
        if len(self._parts()) > 2:
            self._exchange_in_parts(self.local.dist_trial_dual_begin_part, step_size, primal_weight, theta)
        else:
            self.local.dist_trial_dual_begin(step_size, primal_weight, theta)
            self.comm.all_reduce_sum(self.local.exchange_tensor())
        return self.local.dist_trial_end()

    def add_current_primal_to_average(self, weight):
        self.local.add_current_primal_to_average(weight)

    def _refresh_dual_product(self):
        self.local.dist_dual_product_begin()
        self.comm.all_reduce_sum(self.local.exchange_tensor())
        self.local.dist_dual_product_end()

    # ---- average / restart ----
    def average_info(self):
        return self.local.average_info()

    def get_average(self):
        xa, ya = self.local.get_average()
        return xa, self.comm.all_gather_host(ya, self.sizes)

    def reset_average(self):
        self.local.reset_average()

    def restart_to_average(self):
        self.local.restart_to_average()
        self._refresh_dual_product()

    # ---- iterate I/O ----
    def get_current(self):
        x, y = self.local.get_current()
        return x, self.comm.all_gather_host(y, self.sizes)

    def get_dual_product(self):
        return self.local.get_dual_product()

    def set_current(self, x=None, y=None):
        self.local.set_current(x, None if y is None else y[self.lo:self.hi])
        self._refresh_dual_product()

    # ---- standalone mat-vecs for the evaluation branch (host vectors) ----
    def spmv(self, x):
        """A*x: every rank multiplies its row block, slices are concatenated."""
        return self.comm.all_gather_host(self.local.spmv(x), self.sizes)

    def spmv_t(self, y):
        """A'*y = sum_p A_p' y_p: local partial, then a sum over ranks."""
        return self.comm.all_reduce_host(self.local.spmv_t(y[self.lo:self.hi]))

    def close(self):
        self.local.close()


class HipRowShardEngine:
    """HipPdhgEngine + torch view of its exchange buffer, on torch's current
    stream so RCCL and the kernels are ordered by the stream."""

    def __init__(self, problem, lo, hi, device_id):
        import torch
        from .engine import HipPdhgEngine
        torch.cuda.set_device(device_id)
        # A dedicated (non-null) torch stream, made current: the library runs
        # on it and torch.distributed orders RCCL against it.
        self._stream = torch.cuda.Stream(device=device_id)
        torch.cuda.set_stream(self._stream)
        self._eng = HipPdhgEngine(device_id=device_id,
                                  stream=self._stream.cuda_stream,
                                  **shard_rows(problem, lo, hi))
        self._cache = {}
        self.n, self.m = self._eng.n, self._eng.m

    def exchange_tensor(self):
        return hip_exchange_tensor(self._eng, self._cache)

    def __getattr__(self, name):
        return getattr(self._eng, name)


def make_row_partitioned_hip_engine(problem, device_id=None, group=None):
    """Build this rank's shard on its GPU (LOCAL_RANK) and wrap it."""
    import os
    comm = TorchComm(group)
    ranges = partition_rows(problem.constraint_matrix, comm.world_size)
    if device_id is None:
        device_id = int(os.environ.get("LOCAL_RANK", comm.rank))
    lo, hi = ranges[comm.rank]
    local = HipRowShardEngine(problem, lo, hi, device_id)
    return RowPartitionedEngine(local, comm, ranges)
