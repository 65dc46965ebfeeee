/*
 * pdhg_hip.h -- C ABI of the MI355X (gfx950) PDHG inner-step library.
 *
 * This is the drop-in boundary for the one hot path of
 * google-research/FirstOrderLp.jl: everything `take_step` does per iteration
 * (src/primal_dual_hybrid_gradient.jl:442-549, 555-767, called from
 * `optimize` at :1044) plus the state accessors `optimize`'s evaluation /
 * restart branch needs (:892-1023).  The reference has no FFI of its own (it
 * is pure Julia); each entry point below names the reference lines it
 * replaces.  A Julia `ccall` shim and a Python `ctypes` binding over exactly
 * these symbols are shown in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns int: 0 = ok, <0 = invalid argument / unsupported,
 *    >0 = hipError_t from the runtime.  `pdhg_last_error()` gives the text.
 *  - the library owns all device memory.  Host arrays passed in are read
 *    during the call only; output arrays are caller-allocated host memory.
 *  - one host thread per handle; calls return after the handle's stream has
 *    drained whenever they produce host-visible results.
 *  - fp64 throughout; kernels are built with -ffp-contract=off so elementwise
 *    updates round exactly like the reference's unfused Julia broadcasts.
 *
 * Environment variables (read when a handle is created unless noted).  These 27 are the
 * library's run-time knobs; every other PDHG_* name in the sources is a development
 * variable (tuning constants, negative-result paths, fault injection) and is IGNORED unless
 * PDHG_DEV=1 is set as well (tests/conftest.py and tools/ set it).
 *
 *   name                 values (default first)        effect
 *   PDHG_SPMV            auto | stream | tiled         force the product layout: CSR row blocks / L2-tiled sweep
 *   PDHG_SJ              auto | 0 | 1                  sliced jagged copy of stream-class matrices (csrc/sj_kernels.hpp); any other value = auto
 *   PDHG_TUNE            1 | 0                         settle the sweep's chunk variant / XCD dealing by timing the product at create
 *                                                      (0: the static rules stand -- reproducible kernel names and create times;
 *                                                      the bits are the same either way; pdhg_layout_describe reports the choice)
 *   PDHG_SLABS           auto | 0 | 1 | 2              column-slab passes of the stream layout (auto: not for banded / block-local rows; 2: also there)
 *   PDHG_SLAB_MB         4                             slab size in MiB
 *   PDHG_TILE_COLS       auto | <columns>              tile width of the sweep
 *   PDHG_ROW_ORDER       relaxed | strict              rows of > 256 entries summed by their wave / strictly left to right
 *   PDHG_XCD_REMAP       1 | 0                         every XCD walks a contiguous eighth of the row blocks
 *   PDHG_DEVICE_LAYOUT   1 | 0                         build the layouts on the device / on the host
 *   PDHG_HOST_THREADS    min(16, cores) | <n>          host threads of the layout builders
 *   PDHG_MAX_SHARD_NNZ   2^31 - 2 | <n>                entries per row segment (64-bit extents; lowered by tests)
 *   PDHG_LAZY_ACCEPT     1 | 0                         the average's update rides on the next trial's kernels
 *   PDHG_GRAPH           1 | 0                         one HIP-graph launch per trial / separate launches (0 also turns
 *                                                      the persistent trial kernels off)
 *   PDHG_COOP            1 | 0                         one persistent kernel per trial (csrc/trial_kernel.hpp)
 *   PDHG_DEVICE_LOOP     1 | 0                         several take_steps per launch of the persistent kernel
 *   PDHG_SMALL_LP        1 | 0                         whole batches of steps in one workgroup (csrc/small_lp_kernel.hpp)
 *   PDHG_TR_COOP         1 | 0                         a trust-region search as one persistent launch
 *   PDHG_COOP_TRACE      0 | 1                         phase-boundary clock stamps in the persistent kernels (pdhg_trial_timeline)
 *   PDHG_RCCL_LIB        (unset) | <path>              the RCCL library to bind at run time
 *   PDHG_COMM            auto | p2p                    peer kernels instead of RCCL inside one process
 *   PDHG_DIST_OVERLAP    auto | 0 | 1                  per-slice reductions overlapped with the A_p' product
 *   PDHG_DIST_AG_OVERLAP 0 | 1 | 2                     shard groups: xbar's all-gather in column chunks on the comm streams, A_p xbar as one
 *                                                      carried pass per chunk beside it (2: the same passes behind ONE all-gather --
 *                                                      the same bits, nothing overlapped; csrc/dist.hpp)
 *   PDHG_SHARD_THREADS   1 | 0                         one issuing host thread per local shard
 *   PDHG_GROUP_COOP      auto | 0 | 1                  persistent group kernels (1: also across devices)
 *   PDHG_ROCTX           0 | 1                         roctx ranges around entry points and products
 *   PDHG_VERBOSE         0 | 1                         layout / path decisions on stderr
 *   PDHG_DEV             0 | 1                         honour the development variables
 */
#ifndef PDHG_HIP_H_
#define PDHG_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pdhg_handle pdhg_handle;

/* Text of the most recent error on this thread ("" if none). */
const char *pdhg_last_error(void);

/* Library/ABI version (bumped on any signature change). */
int pdhg_abi_version(void);

/*
 * Ingest the (already rescaled) LP exactly as Julia stores it:
 * `constraint_matrix::SparseMatrixCSC{Float64,Int64}` = (colptr[n+1],
 * rowval[nnz], nzval[nnz]) with `index_base` 1 (Julia) or 0 (scipy), plus
 * objective_vector c[n], right_hand_side b[m], variable bounds lb/ub[n]
 * (+-Inf allowed) and num_equalities (rows 0..num_equalities-1 are
 * equalities).  Builds device CSR(A) and CSR(A') with 32-bit indices and
 * zero-initialises x, y, A'y and the weighted-average sums.
 * Replaces: QuadraticProgrammingProblem (src/quadratic_programming.jl:34-76)
 * as the data contract, and the PdhgSolverState zeros(...) construction
 * (src/primal_dual_hybrid_gradient.jl:805-819).
 * `stream`: a hipStream_t to run on (e.g. the caller's torch stream), or NULL
 * for a private stream.  `device_id` < 0 keeps the current device.
 * Sizes: m, n and m + n below 2^31.  nnz may exceed 2^31 - 1 (the reference's
 * index type is Int64): both device copies are then held as SEGMENTS of whole rows
 * inside this one handle (CSR(A) cut by rows, CSR(A') by columns; an entry's address
 * is the segment's base pointer + a 32-bit local offset).  No exchange, every row sum
 * in its reference order: products and trajectories are those of the matrix in one
 * piece.  A single row or column beyond the limit is refused with -2, naming it.
 * (PDHG_HUGE=shards keeps the earlier form: row shards on the same device behind a
 * group handle, peer-kernel exchange, `stream` must be NULL.)
 */
int pdhg_create(pdhg_handle **out, int64_t m, int64_t n, int64_t nnz,
                const int64_t *colptr, const int64_t *rowval,
                const double *nzval, int index_base, const double *c,
                const double *b, const double *lb, const double *ub,
                int64_t num_equalities, int device_id, void *stream);

/*
 * Attach objective_matrix (QP term, CSC n x n) -- src/quadratic_programming.jl:49.
 * Without this call the problem is an LP (objective_matrix == 0).
 */
int pdhg_set_objective_matrix(pdhg_handle *h, int64_t q_nnz,
                              const int64_t *q_colptr, const int64_t *q_rowval,
                              const double *q_nzval, int index_base);

void pdhg_destroy(pdhg_handle *h);

/*
 * One trial step into shadow buffers x', y', A'y' (nothing is committed):
 *   x'  = proj_[lb,ub](x - (step/pw) * (Qx + c - A'y))   compute_next_primal_solution  pdhg.jl:442-470
 *   xb  = x' + theta*(x' - x)                                                           pdhg.jl:486-487
 *   y'  = proj(y + (pw*step) * (b - A*xb))                compute_next_dual_solution    pdhg.jl:472-494
 *   A'y'                                                                                pdhg.jl:492
 * out[0] = dx . (A'y' - A'y)   out[1] = sum dx^2   out[2] = sum dy^2
 * out[3] = sum (A'y' - A'y)^2  out[4] = 0.5 * dx' Q dx (0 for an LP)
 * -- the raw sums compute_interaction_and_movement (pdhg.jl:527-549) and the
 * Malitsky-Pock test (pdhg.jl:615-616) are built from; the host applies
 * abs / sqrt / primal_weight.
 */
int pdhg_trial_step(pdhg_handle *h, double step_size, double primal_weight,
                    double theta, double out[5]);

/* Malitsky-Pock split (pdhg.jl:572-616): x' once, then repeated dual trials. */
int pdhg_trial_primal(pdhg_handle *h, double step_size, double primal_weight);
int pdhg_trial_dual(pdhg_handle *h, double step_size, double primal_weight,
                    double theta, double out[5]);

/*
 * Commit the last trial: x<-x', y<-y', A'y<-A'y' (buffer swap) and
 * sum_x += w*x, sum_y += w*y, counts += 1, weights += w.
 * Replaces update_solution_in_solver_state (pdhg.jl:500-519) +
 * add_to_solution_weighted_average (saddle_point.jl:252-294).  The caller
 * passes the weight (reference quirk: it is solver_state.step_size on entry to
 * take_step, pdhg.jl:512).
 * The two sums are updated lazily: the call itself launches nothing, the next
 * trial's kernels add w*x and w*y where they read x and y anyway, and every other
 * entry point that reads or changes x, y or the sums settles a pending update
 * first -- observable state is always as if the update had happened here, bit for
 * bit (PDHG_LAZY_ACCEPT=0: update in this call with its own kernel).
 */
int pdhg_accept(pdhg_handle *h, double avg_weight);

/*
 * take_step(::AdaptiveStepsizeParams, ...) (pdhg.jl:653-731) in one call: the retry
 * loop around pdhg_trial_step, the scalar step-size rule (pdhg.jl:691-729, on the
 * host, in C) and pdhg_accept.  In/out: *step_size, *total_number_iterations,
 * *cumulative_kkt_passes (the solver-state scalars the reference updates);
 * *numerical_error is set when movement == 0 (pdhg.jl:692-697).  Equivalent to
 * driving pdhg_trial_step / pdhg_accept from the host language statement by statement
 * (julia/FirstOrderLpHIP.jl does that); it exists because an interpreted host spends
 * as long between two calls as a small LP's kernels take.
 */
int pdhg_take_step_adaptive(pdhg_handle *h, double reduction_exponent, double growth_exponent,
                            double *step_size, double primal_weight,
                            int64_t *total_number_iterations, double *cumulative_kkt_passes,
                            int *numerical_error);

/*
 * `n_steps` consecutive pdhg_take_step_adaptive calls: what optimize() runs between two
 * termination evaluations (pdhg.jl:862-1046 does nothing but take_step on those
 * iterations).  Returns after the step that set *numerical_error (the reference notices
 * it at the top of the next iteration); *steps_done = take_steps completed.
 */
int pdhg_take_steps_adaptive(pdhg_handle *h, int64_t n_steps, double reduction_exponent,
                             double growth_exponent, double *step_size, double primal_weight,
                             int64_t *total_number_iterations, double *cumulative_kkt_passes,
                             int *numerical_error, int64_t *steps_done);

/* add_to_primal_solution_weighted_average on the CURRENT x (pdhg.jl:621-627). */
int pdhg_add_current_primal_to_average(pdhg_handle *h, double weight);

/* SolutionWeightedAverage bookkeeping (saddle_point.jl:215-222). */
int pdhg_get_average_info(pdhg_handle *h, int64_t counts[2], double weights[2]);
/* compute_average (saddle_point.jl:296-301): sum / weight.  NULL skips one. */
int pdhg_get_average(pdhg_handle *h, double *x_avg, double *y_avg);
/* reset_solution_weighted_average (saddle_point.jl:238-250). */
int pdhg_reset_average(pdhg_handle *h);
/* current .= avg (saddle_point.jl:808-809) + A'y recompute (pdhg.jl:1018-1022). */
int pdhg_restart_to_average(pdhg_handle *h);

/* Iterate I/O for the host-side evaluation / restart branch (pdhg.jl:892-1023). */
int pdhg_get_current(pdhg_handle *h, double *x, double *y, double *aty);
/* Overwrite x,y (NULL keeps one) and recompute the cached A'y. */
int pdhg_set_current(pdhg_handle *h, const double *x, const double *y);
/* Shadow (trial) buffers; after pdhg_accept they hold the PREVIOUS iterate. */
int pdhg_get_trial(pdhg_handle *h, double *x_next, double *y_next,
                   double *aty_next);

/* Standalone primitives on host vectors: out = A*x (saddle_point.jl:1106),
 * out = A'*y (pdhg.jl:492), for the evaluation branch
 * (iteration_stats_utils.jl:34,162). */
int pdhg_spmv(pdhg_handle *h, const double *x, double *out);
int pdhg_spmv_t(pdhg_handle *h, const double *y, double *out);

/*
 * ---- row-partitioned multi-GPU form, owned by the library --------------------
 * The constraint matrix is 1-D row-partitioned (contiguous row ranges balanced by
 * nonzeros, equalities-first order kept) over `world` GPUs of one node; rank p
 * holds A_p in both layouts, its rows of y / b / sum_y, and OWNS the column slice
 * [p*S, (p+1)*S) of every n-vector.  Per trial step, inside the library:
 *   primal step on the owned slice  ->  all-gather xbar (RCCL over xGMI)
 *   -> y'_p = proj(y_p + sigma (b_p - A_p xbar))  ->  t_p = A_p' y'_p
 *   -> reduce-scatter(sum) t_p = the owned slice of A'y'
 *   -> interaction / movement partial sums on the slice; the scalars of all ranks
 *      are gathered and added in rank order on every rank (identical decisions).
 * Both creators take the GLOBAL problem, exactly like pdhg_create, on every rank;
 * the library partitions the rows itself and keeps only its shard.  The handle
 * they return is used with EVERY other entry point of this header unchanged, and
 * all vector arguments keep their GLOBAL lengths (m, n): trial steps, accept,
 * averages, restarts, evaluation, trust-region bounds and rescaling run sharded
 * behind the same calls; only scalars (and, on request, whole solutions) cross
 * the boundary.  QPs: the objective matrix is replicated (it acts on full
 * n-vectors), so x' is all-gathered as well.
 * No reference counterpart (the reference is single-process); the arithmetic
 * being distributed is src/primal_dual_hybrid_gradient.jl:442-549.
 */
#define PDHG_UNIQUE_ID_BYTES 128
/* One process per GPU (torch.distributed.run, MPI, Julia Distributed ...): rank 0
 * obtains an id, the host sends the 128 bytes to every rank by its own means, and
 * every rank calls pdhg_create_dist with its rank and GPU.  Collective call. */
int pdhg_dist_get_unique_id(void *id /* PDHG_UNIQUE_ID_BYTES */);
int pdhg_create_dist(pdhg_handle **out, int64_t m, int64_t n, int64_t nnz,
                     const int64_t *colptr, const int64_t *rowval,
                     const double *nzval, int index_base, const double *c,
                     const double *b, const double *lb, const double *ub,
                     int64_t num_equalities, int device_id, void *stream,
                     const void *unique_id, int rank, int world);
/*
 * The same with RANK-LOCAL ingest: every rank hands over only ITS rows.  row_bounds[world+1]
 * is the global row partition (ascending, row_bounds[0] = 0, row_bounds[world] = m_global;
 * rank r owns rows row_bounds[r] .. row_bounds[r+1]); (colptr, rowval, nzval) is the CSC of
 * those rows -- all n columns, row indices REBASED to 0 (+ index_base) -- and b_local their
 * right-hand sides.  c, lb, ub are the global n-vectors; num_equalities is global
 * (equalities-first order: rank r's equalities are the rows below num_equalities).  Nothing
 * of the other ranks' rows is read, so a host can generate or load the matrix once and hand
 * each rank its slice.  pdhg_partition_rows gives the library's own nnz-balanced partition of
 * a global matrix (what pdhg_create_dist / pdhg_create_multi use); any contiguous partition
 * is accepted.  Host-only helper: needs no GPU.
 */
int pdhg_partition_rows(int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval,
                        int index_base, int world, int64_t *row_bounds /* world + 1 */);
int pdhg_create_dist_rows(pdhg_handle **out, int64_t m_global, int64_t n,
                          const int64_t *row_bounds, int64_t local_nnz,
                          const int64_t *colptr, const int64_t *rowval, const double *nzval,
                          int index_base, const double *c, const double *b_local,
                          const double *lb, const double *ub, int64_t num_equalities,
                          int device_id, void *stream, const void *unique_id, int rank,
                          int world);
/* One process driving n_devices GPUs (what a single Julia process would use):
 * all ranks live inside the returned handle; every call fans out over the
 * devices (one stream and one RCCL communicator per device, ncclCommInitAll).
 * When device ids repeat (several shards on one GPU) or PDHG_COMM=p2p is set,
 * the exchange uses direct peer-memory kernels instead of RCCL (fixed rank-order
 * sums). */
int pdhg_create_multi(pdhg_handle **out, int64_t m, int64_t n, int64_t nnz,
                      const int64_t *colptr, const int64_t *rowval,
                      const double *nzval, int index_base, const double *c,
                      const double *b, const double *lb, const double *ub,
                      int64_t num_equalities, int n_devices, const int *device_ids);
/* info[0] world size, [1] ranks inside this handle, [2] rank of the handle's first
 * shard, [3] exchange back end (0 RCCL, 1 peer kernels, -1 none), [4],[5] its row
 * range, [6],[7] its owned column range. */
int pdhg_dist_info(pdhg_handle *h, int64_t info[8]);
/*
 * RCCL is bound at RUN time, on the first multi-GPU entry point (csrc/rccl_loader.hpp):
 * PDHG_RCCL_LIB if set, else the librccl the process has already loaded (e.g. torch's), else
 * $ROCM_PATH/lib/librccl.so.1, else the loader's search path.  Its ncclGetVersion() must have
 * the MAJOR version of the rccl.h this library was compiled against, or every multi-GPU
 * creator fails with 2999 and a message.  This call forces the binding and reports it:
 * compiled_version = NCCL_VERSION_CODE of the header, runtime_version = ncclGetVersion(),
 * path = file the symbols came from.  Returns 2999 if RCCL is unavailable or refused.
 */
int pdhg_rccl_info(int *compiled_version, int *runtime_version, char *path, int path_len);
/*
 * Host-side cost of the trial steps so far.  Group handles: seconds until the last launch /
 * collective call of a trial returned (max over the issuing threads: pdhg_create_multi
 * issues every shard's sequence from its own host thread, PDHG_SHARD_THREADS=0 from the
 * calling thread alone) and seconds from then until the scalars were on the host.  Plain
 * handles on the one-launch path: node updates + launch call, and the wait for the result.
 */
int pdhg_host_issue_stats(pdhg_handle *h, int64_t *trials, double *issue_seconds,
                          double *wait_seconds);

/*
 * ---- evaluation branch on the device ("next" row N1) -----------------------
 * Everything optimize()'s evaluation/restart branch (pdhg.jl:892-1023) needs,
 * reduced to scalars on the device so that only scalars cross the boundary.
 * LPs and QPs (for a QP the objective matrix set by pdhg_set_objective_matrix
 * contributes Q x to the gradient and 0.5 x'Qx to the objectives).
 */
enum { PDHG_POINT_CURRENT = 0, PDHG_POINT_AVERAGE = 1, PDHG_POINT_RESTART = 2 };

/* Rescaling vectors (ScaledQpProblem, quadratic_programming.jl:293-298) and the
 * ORIGINAL problem's vectors, for statistics on the unscaled point
 * (evaluate_unscaled_iteration_stats, iteration_stats_utils.jl:413-451). */
int pdhg_set_original_problem(pdhg_handle *h, const double *constraint_rescaling,
                              const double *variable_rescaling, const double *c_o,
                              const double *b_o, const double *lb_o, const double *ub_o);

/*
 * Raw sums/maxes behind compute_convergence_information and
 * compute_infeasibility_information (iteration_stats_utils.jl:228-349) at the
 * unscaled point x_o = x ./ D, y_o = y ./ E:
 *  rows  out[0] sum viol^2   out[1] sum y_o^2   out[2] b_o.y_o   out[3] sum max(-y_o,0)^2 (ineq)
 *        out[4] max|viol|    out[5] max|viol| with b=0 (ray)      out[6] max|y_o|  out[7] max max(-y_o,0)
 *  cols  out[8] sum (g-rc)^2 out[9] sum bound*rc  out[10] sum x_o^2  out[11] c_o.x_o
 *        out[12] sum bound-violation^2   out[13] sum bound*rc (c=0)
 *        out[14] max|g-rc|   out[15] max|x_o|   out[16] max bound violation
 *        out[17] max|g-rc| (c=0)  out[18] max|rc| (c=0)  out[19] max ray bound violation
 *        out[20] x_o.(Q_o x_o)   out[21] max|Q_o x_o|   (both 0 for an LP; out[22..23] reserved)
 * with g = Q_o x_o + c_o - A_o'y_o and rc the reduced costs (iteration_stats_utils.jl:128-148).
 * On one handle the same reduction also carries what a check asks for next -- pdhg_distance_to_restart of the average and
 * of the current iterate, pdhg_point_sumsq of `point` -- and those calls answer from it (the values their own launches
 * produce, bit for bit) until the iterates, the sums or the restart point change.
 */
int pdhg_eval_point(pdhg_handle *h, int point, double out[24]);

/* last_restart_info.{primal,dual}_solution .= current (saddle_point.jl:921-922). */
int pdhg_save_restart_point(pdhg_handle *h);
/* out[0] = sum (x - x_restart)^2, out[1] = sum (y - y_restart)^2 at `point`
 * (weighted_norm distances, saddle_point.jl:445-477, 911-920). */
int pdhg_distance_to_restart(pdhg_handle *h, int point, double out[2]);
/* out[0] = sum x^2, out[1] = sum y^2 of the (scaled) point: the weighted_norm
 * of update_objective_bound_estimates (saddle_point.jl:1024-1027). */
int pdhg_point_sumsq(pdhg_handle *h, int point, double out[2]);
/* Copy one of the three points to the host (NULL skips a vector). */
int pdhg_get_point(pdhg_handle *h, int point, double *x, double *y);

/*
 * bound_optimal_objective (trust_region_utils.jl:271-360) on the scaled LP at
 * `point`, with uniform norm weights per block (define_norms, pdhg.jl:265-277).
 * range: 0 = joint ball (EUCLIDEAN_NORM), 1 = primal block only, 2 = dual block
 * only (the two halves of MAX_NORM).  out[0] = Lagrangian value minus
 * objective_constant, out[1] = sum g_x.(x_tr - x), out[2] = sum g_y.(y_tr - y)
 * (lower bound = L + out[1], upper bound = L - out[2]), out[3] = sum x^2,
 * out[4] = sum y^2, out[5] = t*, out[6] = probe passes of the breakpoint search
 * (after the set-up pass; 0 when every finite breakpoint lies inside the ball).
 * The two value sums are evaluated as sum g d min(t*, breakpoint) -- the same numbers as
 * the reference's sum g (clamp(z + t* d) - z) up to rounding -- so that they come out of
 * the search's own passes.
 * How a call runs (same results to rounding, chosen per handle): n + m <= 4096 in one workgroup; single handles up to
 * n + m = 1M as ONE persistent launch with a grid barrier per probe pass (csrc/tr_coop_kernel.hpp; PDHG_TR_COOP=0 off);
 * otherwise, and on shard groups, a kernel pair and a host round trip per pass.
 */
int pdhg_trust_region_bound(pdhg_handle *h, int point, double primal_weight_norm,
                            double dual_weight_norm, double radius, int range,
                            int approximate, double out[8]);

/*
 * `count` (1..3) such problems in one call: points[p], radii[p], ranges[p] -> out[8 p .. 8 p + 7], each exactly what
 * pdhg_trust_region_bound returns for it (same statements, same grouping of the sums: the same bits as the one-launch
 * form of the single call).  A restart check needs three bounds -- at the average, at the current iterate, at the last
 * restart point (saddle_point.jl:432-496, 551-596) -- and a recorded iteration two more (the halves of MAX_NORM); on
 * single handles of medium size they run as ONE persistent launch whose probe passes share the grid barriers
 * (csrc/tr_coop_kernel.hpp: tr_coop_batch_kernel): the call lasts as long as the problem with the most passes, not the sum.
 * Elsewhere (small problems, shard groups, profiling) it is `count` single calls.  (abi 10)
 */
int pdhg_trust_region_bounds(pdhg_handle *h, int count, const int *points, double primal_weight_norm,
                             double dual_weight_norm, const double *radii, const int *ranges,
                             int approximate, double *out /* 8 * count */);

/*
 * ---- rescaling on the device ("next" row N2) --------------------------------
 * rescale_problem (src/preprocess.jl:631-687) in place on a handle created from
 * the ORIGINAL problem: l_inf_ruiz_iterations of ruiz_rescaling with p = Inf
 * (:412-477), then l2_norm_rescaling (:358-372) if requested, then
 * pock_chambolle_rescaling(alpha) (:508-539) if requested.  Every resident copy
 * of the matrix is scaled entry by entry as (a * (1/e_i)) * (1/d_j), and c, b,
 * lb, ub as scale_problem does (:555-573).  Outputs the cumulative
 * constraint_rescaling[m] / variable_rescaling[n] (ScaledQpProblem,
 * src/quadratic_programming.jl:293-298).  With an objective matrix the Ruiz
 * column factors take the max over the columns of [A; Q] (:425-433) and both
 * resident copies of Q become (D^-1 Q) D^-1 (:562-564).
 */
int pdhg_rescale(pdhg_handle *h, int l_inf_ruiz_iterations, int l2_norm_rescaling,
                 int use_pock_chambolle, double pock_chambolle_alpha,
                 double *constraint_rescaling_out, double *variable_rescaling_out);
/* The handle's (possibly rescaled) objective_vector, right_hand_side and bounds. */
int pdhg_get_problem_vectors(pdhg_handle *h, double *c, double *b, double *lb, double *ub);
/* norm(constraint_matrix, Inf) = max |A_ij| of the resident matrix: the initial
 * step size 1/norm(A, Inf) (src/primal_dual_hybrid_gradient.jl:823-826). */
int pdhg_matrix_max_abs(pdhg_handle *h, double *out);

/* ---- measurement ---------------------------------------------------------- */
enum {
  PDHG_K_PRIMAL = 0,     /* x', xb elementwise                         */
  PDHG_K_SPMV_DUAL = 1,  /* CSR(A)  SpMV + dual update epilogue        */
  PDHG_K_SPMV_ATY = 2,   /* CSR(A') SpMV + interaction epilogue        */
  PDHG_K_FINAL = 3,      /* second-stage reduction of block partials   */
  PDHG_K_ACCEPT = 4,     /* weighted-average AXPY                      */
  PDHG_K_ALLGATHER = 5,       /* group: all-gather of xbar (and x' for a QP)          */
  PDHG_K_REDUCE_SCATTER = 6,  /* group: reduce-scatter of the partials A_p' y'_p       */
  PDHG_K_INTERACTION = 7,     /* group: interaction/movement sums on the owned slice   */
  PDHG_K_COUNT = 8
};
/* Bracket every launch of the hot kernels with hipEvents on the handle's
 * stream and accumulate per-kernel time (profiling mode serialises launches;
 * do not enable inside a throughput-timed region). */
int pdhg_profile_enable(pdhg_handle *h, int enable);
int pdhg_profile_read(pdhg_handle *h, int kernel_id, int64_t *launches,
                      double *total_ms);
/* Algorithmic HBM bytes one launch of `kernel_id` must move. */
int64_t pdhg_kernel_algorithmic_bytes(pdhg_handle *h, int kernel_id);
/* The kernel(s) `kernel_id` stands for ON THIS HANDLE (the SpMV layout is chosen per
 * matrix at create), spelled as rocprofv3 prints them -- template arguments included --
 * and joined by " + " when a product is a group of launches (column-slab passes, the
 * long-row pair): summing those names in a kernel trace of the separate launches gives the
 * time pdhg_profile_read brackets.  h == NULL gives the stream-layout names.  The string is
 * valid until the calling thread's next call. */
const char *pdhg_kernel_name(pdhg_handle *h, int kernel_id);
/* Layout statistics (diagnostics): [0..3] CSR(A) {row blocks, long rows, long
 * chunks, max row nnz}, [4..7] same for CSR(A'), [8],[9] tiled-sweep waves of
 * A / A' (0 = stream layout), [10],[11] their tile widths in columns, [12],[13]
 * column-slab passes of A / A' (0 = single pass), [14] how pdhg_trial_step is launched: 2 one
 * persistent kernel per trial, 1 one graph launch (small / medium LPs; not while
 * profiling), 0 separate launches.  A handle leaves 2 for good when a grid barrier of
 * the persistent kernel times out (device shared with another persistent kernel): that
 * trial is repeated on the other path, with a line on stderr, [15] bit 0 / bit 1: A / A' use
 * equal-nonzero tiles of different widths (skewed columns; [10],[11] are then nominal); bit 2:
 * a small LP -- pdhg_take_steps_adaptive takes its batches in one workgroup with the vectors in
 * LDS (csrc/small_lp_kernel.hpp; pdhg_trial_step itself is launched as [14] says); bit 3: it takes
 * them with the multi-step persistent kernel (several take_steps per launch: small grids). */
int pdhg_layout_info(pdhg_handle *h, int64_t info[16]);
/* The resident layouts of A, A' (and Q, Q') with every choice pdhg_create made for them, as one JSON
 * object: the product kernels, row blocks / column slabs / long rows, the sweep's tile width, chunk
 * variant and XCD dealing WITH the candidates pdhg_create timed on the matrix and their times
 * ("chosen_by": "timing at create" | "static rule"), the sliced jagged copy's window, hub threshold,
 * hub rows and fill.  A handle's dispatch -- hence its timing, never its bits -- can depend on a
 * measurement taken at create (PDHG_TUNE=0 pins the static rules): a bench line or a profile quotes
 * this text to say which variant it ran.  Writes at most cap - 1 characters + NUL into buf (buf may be
 * NULL); returns the full length of the text, or < 0 on error.  (abi 11) */
int pdhg_layout_describe(pdhg_handle *h, char *buf, int cap);
/* Diagnostics: order-sensitive 64-bit checksums of every device array of the two layouts
 * (out[0..16) CSR(A): row pointers, columns, values, row blocks, the four long-row tables, the
 * sweep's pk / tv / wave rows / entry offsets / step offsets / step tiles / workgroup steps, the
 * plan's scalars; a stream layout's column slabs -- row pointers, columns, values, row blocks --
 * in the first four of the sweep's slots; out[16..32) the same for CSR(A')).  Two handles with equal checksums hold
 * bit-identical layouts: how the device-side layout construction is held to the host builders. */
int pdhg_layout_checksums(pdhg_handle *h, uint64_t out[32]);
/* Measurement only: best-of-`reps` rate of a[i] = b[i] + s*c[i] over `len`
 * doubles (len % 4 == 0) on this handle's device and stream (24*len bytes per pass), in GB/s --
 * the box's own streaming ceiling to put beside the 8 TB/s spec figure. */
int pdhg_measure_triad(pdhg_handle *h, int64_t len, int reps, double *gbps);

/* Measurement aid for the benchmark line's `roofline.ceiling_frac`: the gather rate the tiled sweep's ACCESS PATTERN
 * reaches on this device with nothing else in the kernel (same workgroup geometry, tiles, pacing barriers and entry
 * streams as the product of this handle's constraint matrix; no accumulators, no epilogue).  out[0] G gathers/s of the
 * pattern, out[1] ms per pass, out[2] G gathers/s with every gather inside one resident window and no barriers,
 * out[3] entries per (wave, tile) cell, out[4] waves, out[5] tiles.  No reference counterpart (the reference has no
 * device code); stands beside pdhg_measure_triad. */
int pdhg_measure_sweep_ceiling(pdhg_handle *h, int64_t rows, int64_t cols, int64_t nnz, int reps, double out[6]);

/* Measurement aid for the benchmark line's `trial_timeline`: where the time of one trial goes INSIDE the persistent
 * trial kernels (PDHG_COOP_TRACE=1 makes them stamp the 100 MHz wall clock at every phase boundary, per workgroup).
 * out[0..4] mean duration in us over the workgroups of phase 0 (x', xbar), barrier 1, phase 1 (A xbar, y'), barrier 2,
 * phase 2 (A'y', sums); out[5..9] the slowest workgroup's; out[10] last workgroup out of phase 2; out[11] barrier 3's
 * global phase complete (multi-step kernel, else 0); out[12] decision known / results published; out[13] workgroups.
 * Returns 1 when nothing was traced.  No reference counterpart. */
int pdhg_trial_timeline(pdhg_handle *h, double out[14]);
/* Measurement only: what a HIP-event bracket reports for EMPTY launches on this handle's
 * stream -- out[0] ms for one empty kernel between the two events, out[1] ms for every further
 * launch inside the same bracket (best of `reps`).  pdhg_profile_read's brackets contain this
 * much that is not kernel time: about 10 us for the first launch and 3 us per further one,
 * which matters for the 5-50 us kernels of a small LP (a rocprofv3 kernel trace, which times
 * the kernels themselves, is shorter by that amount). */
int pdhg_measure_launch_overhead(pdhg_handle *h, int reps, double out[2]);

/* Self-test (abi 11, round 6): the block reduction of the evaluation / trust-region kernels shares the wave sums of a
 * workgroup's quantities between the lanes (csrc/eval_kernels.hpp: WaveSplit) instead of running one tree per quantity;
 * both forms sum every quantity over the same balanced tree, so their results must agree BIT FOR BIT.  Runs both on
 * pseudo-random data (seed) for 8, 16, 22, 30 and 64 quantities per lane: out[0] = wave totals compared, out[1] = totals
 * whose bits differ (0 expected).  No reference counterpart: a property of this implementation. */
int pdhg_selftest_wave_sums(pdhg_handle *h, int64_t seed, int64_t out[2]);

#ifdef __cplusplus
}
#endif
#endif /* PDHG_HIP_H_ */
