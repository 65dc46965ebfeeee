"""Device-backed PDHG solver state: the Python binding over the C ABI
(include/pdhg_hip.h).  One ``HipPdhgEngine`` == one ``pdhg_handle``.

The engine owns the n-/m-length vectors of ``PdhgSolverState``
(primal_dual_hybrid_gradient.jl:205-258) and ``SolutionWeightedAverage``
(saddle_point.jl:215-222) on the GPU; the scalars (step_size, primal_weight,
counters) stay with the host driver in primal_dual_hybrid_gradient.py.

No fallback: constructing an engine without the HIP library or without a GPU
raises.
"""
import ctypes

import numpy as np

from . import _lib

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int64)


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _pd(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _pi(a):
    return a.ctypes.data_as(_ip)


class HipPdhgEngine:
    """All arguments describe the *rescaled* problem the iterations run on."""

    def __init__(self, constraint_matrix, objective_vector, right_hand_side,
                 variable_lower_bound, variable_upper_bound, num_equalities,
                 objective_matrix=None, device_id=-1, stream=None,
                 device_ids=None, unique_id=None, rank=None, world=None):
        """One GPU by default.  Row-partitioned over several GPUs, inside the
        library (include/pdhg_hip.h, "row-partitioned multi-GPU form"):

        * ``device_ids=[...]``: this process drives all of them (``pdhg_create_multi``);
        * ``unique_id, rank, world``: one process per GPU (``pdhg_create_dist``),
          ``unique_id`` from ``dist_unique_id()`` on rank 0, sent to all ranks by the host.

        Every rank passes the GLOBAL problem; all methods keep global vector lengths."""
        self._L = _lib.lib()
        A = constraint_matrix
        self.m, self.n = int(A.shape[0]), int(A.shape[1])
        colptr, rowval, nzval = _i(A.indptr), _i(A.indices), _d(A.data)
        c, b = _d(objective_vector), _d(right_hand_side)
        lb, ub = _d(variable_lower_bound), _d(variable_upper_bound)
        if c.shape != (self.n,) or lb.shape != (self.n,) or ub.shape != (self.n,) \
                or b.shape != (self.m,):
            raise ValueError("vector lengths do not match the constraint matrix")
        h = ctypes.c_void_p()
        common = (self.m, self.n, len(nzval), _pi(colptr), _pi(rowval), _pd(nzval), 0,
                  _pd(c), _pd(b), _pd(lb), _pd(ub), int(num_equalities))
        if device_ids is not None:
            ids = (ctypes.c_int * len(device_ids))(*[int(d) for d in device_ids])
            _lib.check(self._L.pdhg_create_multi(ctypes.byref(h), *common, len(device_ids), ids))
        elif unique_id is not None:
            uid = ctypes.create_string_buffer(bytes(unique_id), _lib.UNIQUE_ID_BYTES)
            _lib.check(self._L.pdhg_create_dist(
                ctypes.byref(h), *common, int(device_id),
                ctypes.c_void_p(stream) if stream else None, uid, int(rank), int(world)))
        else:
            _lib.check(self._L.pdhg_create(
                ctypes.byref(h), *common, int(device_id),
                ctypes.c_void_p(stream) if stream else None))
        self._h = h
        if objective_matrix is not None and objective_matrix.nnz > 0:
            Q = objective_matrix
            qc, qr, qv = _i(Q.indptr), _i(Q.indices), _d(Q.data)
            _lib.check(self._L.pdhg_set_objective_matrix(
                self._h, len(qv), _pi(qc), _pi(qr), _pd(qv), 0))

    @classmethod
    def from_row_shard(cls, m_global, row_bounds, constraint_rows, objective_vector, right_hand_side_rows,
                       variable_lower_bound, variable_upper_bound, num_equalities, unique_id, rank, world,
                       device_id=-1, stream=None, objective_matrix=None):
        """One process per GPU with RANK-LOCAL ingest (``pdhg_create_dist_rows``): this rank
        passes only ITS rows -- ``constraint_rows`` = rows row_bounds[rank]..row_bounds[rank+1] of
        the global matrix as an (hi - lo) x n sparse matrix, ``right_hand_side_rows`` their
        right-hand sides -- plus the global n-vectors; ``num_equalities`` is global.  The handle
        behaves exactly like one from ``unique_id=, rank=, world=`` (global vector lengths)."""
        self = cls.__new__(cls)
        self._L = _lib.lib()
        A = constraint_rows.tocsc()
        bounds = _i(row_bounds)
        if bounds.shape != (world + 1,):
            raise ValueError("row_bounds must hold world + 1 entries")
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        self.m, self.n = int(m_global), int(A.shape[1])
        if A.shape[0] != hi - lo:
            raise ValueError("constraint_rows does not match row_bounds[rank]..row_bounds[rank+1]")
        colptr, rowval, nzval = _i(A.indptr), _i(A.indices), _d(A.data)
        c, b = _d(objective_vector), _d(right_hand_side_rows)
        lb, ub = _d(variable_lower_bound), _d(variable_upper_bound)
        if c.shape != (self.n,) or lb.shape != (self.n,) or ub.shape != (self.n,) or b.shape != (hi - lo,):
            raise ValueError("vector lengths do not match the row shard")
        h = ctypes.c_void_p()
        uid = ctypes.create_string_buffer(bytes(unique_id), _lib.UNIQUE_ID_BYTES)
        _lib.check(self._L.pdhg_create_dist_rows(
            ctypes.byref(h), self.m, self.n, _pi(bounds), len(nzval), _pi(colptr), _pi(rowval), _pd(nzval), 0,
            _pd(c), _pd(b), _pd(lb), _pd(ub), int(num_equalities), int(device_id),
            ctypes.c_void_p(stream) if stream else None, uid, int(rank), int(world)))
        self._h = h
        if objective_matrix is not None and objective_matrix.nnz > 0:
            Q = objective_matrix
            qc, qr, qv = _i(Q.indptr), _i(Q.indices), _d(Q.data)
            _lib.check(self._L.pdhg_set_objective_matrix(self._h, len(qv), _pi(qc), _pi(qr), _pd(qv), 0))
        return self

    @staticmethod
    def partition_rows(constraint_matrix, world):
        """The library's nnz-balanced contiguous row partition (``pdhg_partition_rows``; host-only,
        needs no GPU): ``world + 1`` ascending bounds."""
        A = constraint_matrix.tocsc()
        colptr, rowval = _i(A.indptr), _i(A.indices)
        out = np.zeros(world + 1, dtype=np.int64)
        _lib.check(_lib.lib().pdhg_partition_rows(int(A.shape[0]), int(A.shape[1]), _pi(colptr), _pi(rowval), 0,
                                                  int(world), _pi(out)))
        return out

    @staticmethod
    def rccl_info():
        """Which RCCL the library bound at run time: dict(compiled_version, runtime_version, path).
        Raises if RCCL is unavailable or its major version differs from the compiled header's."""
        cv, rv = ctypes.c_int(0), ctypes.c_int(0)
        buf = ctypes.create_string_buffer(1024)
        _lib.check(_lib.lib().pdhg_rccl_info(ctypes.byref(cv), ctypes.byref(rv), buf, 1024))
        return {"compiled_version": cv.value, "runtime_version": rv.value, "path": buf.value.decode()}

    def host_issue_stats(self):
        """(trials, seconds issuing, seconds waiting) of the trial steps so far (``pdhg_host_issue_stats``)."""
        n = ctypes.c_int64(0)
        a, b = ctypes.c_double(0.0), ctypes.c_double(0.0)
        _lib.check(self._L.pdhg_host_issue_stats(self._h, ctypes.byref(n), ctypes.byref(a), ctypes.byref(b)))
        return n.value, a.value, b.value

    @staticmethod
    def dist_unique_id():
        """128 opaque bytes identifying a new RCCL communicator (rank 0 calls this
        and sends the bytes to the other ranks)."""
        buf = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        _lib.check(_lib.lib().pdhg_dist_get_unique_id(buf))
        return buf.raw

    def dist_info(self):
        info = np.zeros(8, dtype=np.int64)
        _lib.check(self._L.pdhg_dist_info(self._h, _pi(info)))
        keys = ["world", "local_ranks", "rank", "backend", "row_lo", "row_hi", "col_lo", "col_hi"]
        return dict(zip(keys, info.tolist()))

    @classmethod
    def from_problem(cls, problem, **kw):
        return cls(problem.constraint_matrix, problem.objective_vector,
                   problem.right_hand_side, problem.variable_lower_bound,
                   problem.variable_upper_bound, problem.num_equalities,
                   objective_matrix=problem.objective_matrix, **kw)

    def close(self):
        if getattr(self, "_h", None):
            self._L.pdhg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the hot path -------------------------------------------------------
    def trial_step(self, step_size, primal_weight, theta=1.0):
        out = np.empty(5)
        _lib.check(self._L.pdhg_trial_step(self._h, step_size, primal_weight,
                                           theta, _pd(out)))
        return out

    def trial_primal(self, step_size, primal_weight):
        _lib.check(self._L.pdhg_trial_primal(self._h, step_size, primal_weight))

    def trial_dual(self, step_size, primal_weight, theta):
        out = np.empty(5)
        _lib.check(self._L.pdhg_trial_dual(self._h, step_size, primal_weight,
                                           theta, _pd(out)))
        return out

    def accept(self, avg_weight):
        _lib.check(self._L.pdhg_accept(self._h, avg_weight))

    def take_step_adaptive(self, reduction_exponent, growth_exponent, step_size, primal_weight,
                           total_number_iterations, cumulative_kkt_passes):
        """pdhg_take_step_adaptive: returns (step_size, total_number_iterations,
        cumulative_kkt_passes, numerical_error)."""
        ss = ctypes.c_double(step_size)
        it = ctypes.c_int64(total_number_iterations)
        kkt = ctypes.c_double(cumulative_kkt_passes)
        err = ctypes.c_int(0)
        _lib.check(self._L.pdhg_take_step_adaptive(
            self._h, reduction_exponent, growth_exponent, ctypes.byref(ss), primal_weight,
            ctypes.byref(it), ctypes.byref(kkt), ctypes.byref(err)))
        return ss.value, it.value, kkt.value, bool(err.value)

    def take_steps_adaptive(self, n_steps, reduction_exponent, growth_exponent, step_size, primal_weight,
                            total_number_iterations, cumulative_kkt_passes):
        """pdhg_take_steps_adaptive: `n_steps` take_steps in one call.  Returns (step_size,
        total_number_iterations, cumulative_kkt_passes, numerical_error, steps_done)."""
        ss = ctypes.c_double(step_size)
        it = ctypes.c_int64(total_number_iterations)
        kkt = ctypes.c_double(cumulative_kkt_passes)
        err = ctypes.c_int(0)
        done = ctypes.c_int64(0)
        _lib.check(self._L.pdhg_take_steps_adaptive(
            self._h, int(n_steps), reduction_exponent, growth_exponent, ctypes.byref(ss), primal_weight,
            ctypes.byref(it), ctypes.byref(kkt), ctypes.byref(err), ctypes.byref(done)))
        return ss.value, it.value, kkt.value, bool(err.value), done.value

    def add_current_primal_to_average(self, weight):
        _lib.check(self._L.pdhg_add_current_primal_to_average(self._h, weight))

    # ---- weighted average / restarts ---------------------------------------
    def average_info(self):
        counts = np.zeros(2, dtype=np.int64)
        weights = np.zeros(2)
        _lib.check(self._L.pdhg_get_average_info(self._h, _pi(counts),
                                                 _pd(weights)))
        return int(counts[0]), int(counts[1]), float(weights[0]), float(weights[1])

    def get_average(self):
        xa, ya = np.empty(self.n), np.empty(self.m)
        _lib.check(self._L.pdhg_get_average(self._h, _pd(xa), _pd(ya)))
        return xa, ya

    def reset_average(self):
        _lib.check(self._L.pdhg_reset_average(self._h))

    def restart_to_average(self):
        _lib.check(self._L.pdhg_restart_to_average(self._h))

    # ---- iterate I/O ---------------------------------------------------------
    def get_current(self):
        x, y = np.empty(self.n), np.empty(self.m)
        _lib.check(self._L.pdhg_get_current(self._h, _pd(x), _pd(y), None))
        return x, y

    def get_dual_product(self):
        aty = np.empty(self.n)
        _lib.check(self._L.pdhg_get_current(self._h, None, None, _pd(aty)))
        return aty

    def set_current(self, x=None, y=None):
        x = None if x is None else _d(x)
        y = None if y is None else _d(y)
        _lib.check(self._L.pdhg_set_current(self._h, _pd(x), _pd(y)))

    def get_trial(self):
        x, y, a = np.empty(self.n), np.empty(self.m), np.empty(self.n)
        _lib.check(self._L.pdhg_get_trial(self._h, _pd(x), _pd(y), _pd(a)))
        return x, y, a

    # ---- standalone primitives ------------------------------------------------
    def spmv(self, x):
        x = _d(x)
        out = np.empty(self.m)
        _lib.check(self._L.pdhg_spmv(self._h, _pd(x), _pd(out)))
        return out

    def spmv_t(self, y):
        y = _d(y)
        out = np.empty(self.n)
        _lib.check(self._L.pdhg_spmv_t(self._h, _pd(y), _pd(out)))
        return out

    # ---- rescaling on the device --------------------------------------------------
    supports_device_rescaling = True

    def rescale(self, l_inf_ruiz_iterations, l2_norm_rescaling, pock_chambolle_alpha):
        """rescale_problem (preprocess.jl:631-687) in place on the device;
        returns (constraint_rescaling, variable_rescaling)."""
        e, dvec = np.empty(self.m), np.empty(self.n)
        use_pc = pock_chambolle_alpha is not None
        _lib.check(self._L.pdhg_rescale(self._h, int(l_inf_ruiz_iterations),
                                        int(bool(l2_norm_rescaling)), int(use_pc),
                                        float(pock_chambolle_alpha) if use_pc else 0.0,
                                        _pd(e), _pd(dvec)))
        return e, dvec

    def get_problem_vectors(self):
        c, b = np.empty(self.n), np.empty(self.m)
        lb, ub = np.empty(self.n), np.empty(self.n)
        _lib.check(self._L.pdhg_get_problem_vectors(self._h, _pd(c), _pd(b), _pd(lb), _pd(ub)))
        return c, b, lb, ub

    def matrix_max_abs(self):
        out = np.zeros(1)
        _lib.check(self._L.pdhg_matrix_max_abs(self._h, _pd(out)))
        return float(out[0])

    # ---- evaluation branch on the device -------------------------------------------
    supports_device_evaluation = True

    def set_original_problem(self, constraint_rescaling, variable_rescaling, c_o, b_o, lb_o, ub_o):
        arrs = [_d(a) for a in (constraint_rescaling, variable_rescaling, c_o, b_o, lb_o, ub_o)]
        _lib.check(self._L.pdhg_set_original_problem(self._h, *[_pd(a) for a in arrs]))

    def eval_point(self, point):
        out = np.empty(24)
        _lib.check(self._L.pdhg_eval_point(self._h, point, _pd(out)))
        return out

    def save_restart_point(self):
        _lib.check(self._L.pdhg_save_restart_point(self._h))

    def distance_to_restart(self, point):
        out = np.empty(2)
        _lib.check(self._L.pdhg_distance_to_restart(self._h, point, _pd(out)))
        return float(out[0]), float(out[1])

    def point_sumsq(self, point):
        out = np.empty(2)
        _lib.check(self._L.pdhg_point_sumsq(self._h, point, _pd(out)))
        return float(out[0]), float(out[1])

    def get_point(self, point):
        x, y = np.empty(self.n), np.empty(self.m)
        _lib.check(self._L.pdhg_get_point(self._h, point, _pd(x), _pd(y)))
        return x, y

    def trust_region_bound(self, point, primal_weight_norm, dual_weight_norm, radius,
                           norm_range, approximate=False):
        out = np.empty(8)
        _lib.check(self._L.pdhg_trust_region_bound(
            self._h, point, primal_weight_norm, dual_weight_norm, radius, norm_range,
            int(bool(approximate)), _pd(out)))
        return out

    def trust_region_bounds(self, points, primal_weight_norm, dual_weight_norm, radii, norm_ranges, approximate=False):
        """Up to three trust-region problems in one call (pdhg_trust_region_bounds): one row of eight per problem."""
        k = len(points)
        pts = np.ascontiguousarray(points, dtype=np.int32)
        rad = np.ascontiguousarray(radii, dtype=np.float64)
        rng = np.ascontiguousarray(norm_ranges, dtype=np.int32)
        out = np.empty(8 * k)
        _lib.check(self._L.pdhg_trust_region_bounds(
            self._h, k, ctypes.c_void_p(pts.ctypes.data), primal_weight_norm, dual_weight_norm, _pd(rad),
            ctypes.c_void_p(rng.ctypes.data), int(bool(approximate)), _pd(out)))
        return out.reshape(k, 8)

    def measure_triad(self, length=1 << 26, reps=5):
        """GB/s of a = b + s*c over ``length`` doubles on this device (measurement only)."""
        out = ctypes.c_double()
        _lib.check(self._L.pdhg_measure_triad(self._h, int(length), int(reps), ctypes.byref(out)))
        return out.value

    def measure_sweep_ceiling(self, rows, cols, nnz, reps=3):
        """What the tiled sweep's access pattern reaches on this device with nothing else in the kernel
        (pdhg_measure_sweep_ceiling; measurement only): dict with G gathers/s of the pattern, of the all-hit window,
        the pass time and the probe's geometry."""
        out = np.zeros(6)
        _lib.check(self._L.pdhg_measure_sweep_ceiling(self._h, int(rows), int(cols), int(nnz), int(reps), _pd(out)))
        return {"pattern_Ggathers_per_s": float(out[0]), "pattern_ms": float(out[1]), "all_hit_window_Ggathers_per_s": float(out[2]),
                "entries_per_cell": int(out[3]), "waves": int(out[4]), "tiles": int(out[5])}

    def trial_timeline(self):
        """Phase timeline (us) of the last one-launch trial, or None when nothing was traced (PDHG_COOP_TRACE=1)."""
        out = np.zeros(14)
        if self._L.pdhg_trial_timeline(self._h, _pd(out)) != 0:
            return None
        names = ["phase0_primal", "barrier1", "phase1_A_xbar", "barrier2", "phase2_At_y"]
        d = {"mean_us": {n: round(float(out[k]), 2) for k, n in enumerate(names)},
             "max_us": {n: round(float(out[5 + k]), 2) for k, n in enumerate(names)},
             "last_workgroup_out_of_phase2_us": round(float(out[10]), 2), "workgroups": int(out[13])}
        if out[11] > 0:
            d["barrier3_global_phase_complete_us"] = round(float(out[11]), 2)
            d["decision_known_to_last_workgroup_us"] = round(float(out[12]), 2)
        else:
            d["results_published_us"] = round(float(out[12]), 2)
        return d

    def layout_checksums(self):
        """32 order-sensitive checksums of the device arrays of both layouts (pdhg_layout_checksums)."""
        out = np.zeros(32, dtype=np.uint64)
        _lib.check(self._L.pdhg_layout_checksums(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
        return out

    def measure_launch_overhead(self, reps=20):
        """(ms for one empty launch between two HIP events, ms per further launch in the same bracket)."""
        out = np.zeros(2)
        _lib.check(self._L.pdhg_measure_launch_overhead(self._h, int(reps), _pd(out)))
        return float(out[0]), float(out[1])

    def selftest_wave_sums(self, seed=0):
        """pdhg_selftest_wave_sums: (wave totals compared, totals whose bits differ between the shared wave reduction of the
        check kernels and one DPP tree per quantity) -- the second must be 0."""
        out = np.zeros(2, dtype=np.int64)
        _lib.check(self._L.pdhg_selftest_wave_sums(self._h, int(seed), _pi(out)))
        return int(out[0]), int(out[1])

    # ---- measurement -------------------------------------------------------------
    def profile_enable(self, enable=True):
        _lib.check(self._L.pdhg_profile_enable(self._h, int(bool(enable))))

    def profile_read(self, kernel_id):
        cnt = ctypes.c_int64()
        ms = ctypes.c_double()
        _lib.check(self._L.pdhg_profile_read(self._h, kernel_id,
                                             ctypes.byref(cnt), ctypes.byref(ms)))
        return cnt.value, ms.value

    def kernel_algorithmic_bytes(self, kernel_id):
        return int(self._L.pdhg_kernel_algorithmic_bytes(self._h, kernel_id))

    def kernel_name(self, kernel_id):
        return self._L.pdhg_kernel_name(self._h, kernel_id).decode()

    def layout_describe(self):
        """pdhg_layout_describe: the resident layouts and every choice pdhg_create made (incl. those settled by timing)."""
        import ctypes
        import json
        need = self._L.pdhg_layout_describe(self._h, None, 0)
        if need < 0:
            _lib.check(need)
        buf = ctypes.create_string_buffer(need + 1)
        self._L.pdhg_layout_describe(self._h, buf, need + 1)
        return json.loads(buf.value.decode())

    def layout_info(self):
        info = np.zeros(16, dtype=np.int64)
        _lib.check(self._L.pdhg_layout_info(self._h, _pi(info)))
        keys = ["A_blocks", "A_long_rows", "A_long_chunks", "A_max_row_nnz",
                "At_blocks", "At_long_rows", "At_long_chunks", "At_max_row_nnz",
                "A_tiled_waves", "At_tiled_waves", "A_tile_cols", "At_tile_cols",
                "A_slabs", "At_slabs", "trial_graph", "var_tiles"]   # var_tiles: bit 0 = A, bit 1 = A'
        out = dict(zip(keys, info.tolist()))
        for k in ("A", "At"):    # the product kernels run on the sliced jagged layout (csrc/sj_kernels.hpp)
            out[k + "_sj"] = (out[k + "_slabs"] >> 8) & 1
            out[k + "_pipe"] = (out[k + "_slabs"] >> 9) & 1    # the row blocks as a persistent pipelined launch
            out[k + "_slabs"] &= 255
        out["small_lp"] = (out["var_tiles"] >> 2) & 1      # batches of take_steps run in the one-workgroup LDS kernel
        out["device_loop"] = (out["var_tiles"] >> 3) & 1   # ... in the multi-step persistent kernel (small grids)
        out["steps_local"] = (out["var_tiles"] >> 4) & 1   # ... whose workgroups all sit on one XCD (<= 32 row blocks + chunks per product)
        # nnz beyond the 32-bit entry limit: the matrix is held as this many segments of whole rows (0: one piece)
        out["A_segments"] = (out["var_tiles"] >> 8) & 255
        out["At_segments"] = (out["var_tiles"] >> 16) & 255
        # shard groups: trials taken as one persistent kernel per shard (group_kernel.hpp), and its fall-backs
        out["group_coop_trials"] = (out["var_tiles"] >> 24) & 65535
        out["group_coop_fallbacks"] = (out["var_tiles"] >> 40) & 255
        # trust-region calls taken as one persistent launch (tr_coop_kernel.hpp)
        out["tr_coop_calls"] = (out["var_tiles"] >> 48) & 16383
        out["var_tiles"] &= 3
        try:      # the sliced jagged copies of the objective matrix, the form of every copy (pdhg_layout_describe)
            desc = self.layout_describe()
            for k in ("A", "At", "Q", "Qt"):
                if k in desc:
                    sj = desc[k].get("sliced_jagged")
                    if k in ("Q", "Qt"):
                        out[k + "_sj"] = 1 if sj else 0
                    out[k + "_sj_wide"] = 1 if sj and sj["slices_per_wave"] > 1 else 0
                    out[k + "_sj_hub_rows"] = sj["hub_rows"] if sj else 0
        except Exception:      # a diagnostic: never fail layout_info for it
            pass
        for k in ("A", "At"):    # width in bits of an entry's column field
            out[k + "_tile_shift"] = max(1, (out[k + "_tile_cols"] - 1).bit_length()) if out[k + "_tile_cols"] else 0
        return out
