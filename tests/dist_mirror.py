"""Host MIRROR of csrc/dist.hpp for the world_size > 1 CPU tests -- test infrastructure, not product code.

The same algorithm as the library's row-partitioned form -- row shards, owned column slices, reduce-scatter -> slice ->
all-gather, rank-ordered sums, scalars combined in rank order -- restated in numpy over a local engine object (the CPU
oracle in tests/test_distributed_gloo.py) and a small collective interface (torch.distributed / gloo), so that the
N > 1 logic is exercised on a box without GPUs.  It mirrors csrc/dist.hpp step by step and never runs on a GPU;
the product's exchange is issued by the library itself (RCCL or peer kernels).  Moved here from
firstorderlp.jl_amd/distributed.py in round 4 (it was the one piece of test-only code inside the package).
"""
import numpy as np

from firstorderlp_jl_amd.distributed import slice_stride




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
