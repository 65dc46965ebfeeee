# FirstOrderLpHIP.jl -- `ccall` shim over libpdhg_hip.so (include/pdhg_hip.h, abi 9).
#
# Drop-in for FirstOrderLp.jl's PDHG path on MI355X:
#
#     using FirstOrderLp, FirstOrderLpHIP
#     params = FirstOrderLpHIP.HipPdhgParameters(pdhg_params)               # one GPU
#     params = FirstOrderLpHIP.HipPdhgParameters(pdhg_params; devices = 0:7) # row-partitioned over 8 GPUs
#     output = FirstOrderLp.optimize(params, qp)      # same call as scripts/solve_qp.jl:110
#
# `optimize` below is src/primal_dual_hybrid_gradient.jl:782-1049 with every
# n-/m-length vector operation behind the C ABI: the ORIGINAL problem is uploaded
# once (`pdhg_create` / `pdhg_create_multi`), rescaled on the device
# (`pdhg_rescale`), iterated (`pdhg_trial_step` / `pdhg_accept`), and the
# evaluation / restart branch reads scalars only (`pdhg_eval_point`,
# `pdhg_trust_region_bound`, `pdhg_distance_to_restart`, ...).  The step-size rules,
# restart scheme, primal-weight update and termination stay here, on the host, with
# the reference's own arithmetic.  It mirrors firstorderlp.jl_amd/
# primal_dual_hybrid_gradient.py, evaluation.py and saddle_point.py statement by
# statement.  NOT executed in the build environment (no Julia there);
# tests/test_julia_shim.py checks that every export of the header has its ccall here.
module FirstOrderLpHIP

import FirstOrderLp
using LinearAlgebra
using SparseArrays
import Random

const LIB = get(ENV, "PDHG_HIP_LIB", "libpdhg_hip.so")
const ABI_VERSION = 11
const POINT_CURRENT = Cint(0)
const POINT_AVERAGE = Cint(1)
const POINT_RESTART = Cint(2)
const UNIQUE_ID_BYTES = 128

last_error() = unsafe_string(ccall((:pdhg_last_error, LIB), Cstring, ()))
check(rc::Cint) = rc == 0 || error("pdhg_hip error $rc: " * last_error())

function __init__()
  v = ccall((:pdhg_abi_version, LIB), Cint, ())
  v == ABI_VERSION || error("libpdhg_hip.so has abi $v, this shim needs $ABI_VERSION")
end

# ------------------------------------------------------------------ the handle

"Device-backed PdhgSolverState (pdhg.jl:205-258): vectors on the GPU(s), scalars here."
mutable struct HipSolverState
  handle::Ptr{Cvoid}
  primal_size::Int64
  dual_size::Int64
  step_size::Float64
  primal_weight::Float64
  numerical_error::Bool
  cumulative_kkt_passes::Float64
  total_number_iterations::Int64
  required_ratio::Union{Float64,Nothing}
  ratio_step_sizes::Union{Float64,Nothing}
end

# NOTE: ccall's argument-type tuple must be a LITERAL tuple (a splatted constant does not
# lower: `T...` is only legal as the trailing vararg marker), so the twelve common creator
# arguments (m, n, nnz, colptr, rowval, nzval, index_base, c, b, lb, ub, num_equalities) are
# spelled out in each of the three creators below; tests/test_julia_shim.py checks every
# ccall's arity and C types against include/pdhg_hip.h.

"""
Ingest a problem exactly as Julia stores it (SparseMatrixCSC{Float64,Int64},
1-based => index_base = 1).  `devices === nothing`: one GPU (`pdhg_create`, current
device).  `devices = [0, 1, ...]`: this process drives all of them, the constraint
matrix is row-partitioned inside the library (`pdhg_create_multi`).  `dist = (id,
rank, world)`: one process per GPU (`pdhg_create_dist`), `id` from `unique_id()`.
"""
function HipSolverState(problem::FirstOrderLp.QuadraticProgrammingProblem;
                        devices = nothing, dist = nothing, device_id::Integer = -1)
  A = problem.constraint_matrix
  m, n = size(A)
  h = Ref{Ptr{Cvoid}}(C_NULL)
  GC.@preserve A problem begin
    if devices !== nothing
      ids = Cint.(collect(devices))
      check(ccall((:pdhg_create_multi, LIB), Cint,
        (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Cint,
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Cint, Ptr{Cint}),
        h, m, n, length(A.nzval), A.colptr, A.rowval, A.nzval, 1,
        problem.objective_vector, problem.right_hand_side,
        problem.variable_lower_bound, problem.variable_upper_bound,
        problem.num_equalities, length(ids), ids))
    elseif dist !== nothing
      id, rank, world = dist
      check(ccall((:pdhg_create_dist, LIB), Cint,
        (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Cint,
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Cint, Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint),
        h, m, n, length(A.nzval), A.colptr, A.rowval, A.nzval, 1,
        problem.objective_vector, problem.right_hand_side,
        problem.variable_lower_bound, problem.variable_upper_bound,
        problem.num_equalities, device_id, C_NULL, id, rank, world))
    else
      check(ccall((:pdhg_create, LIB), Cint,
        (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Cint,
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Cint, Ptr{Cvoid}),
        h, m, n, length(A.nzval), A.colptr, A.rowval, A.nzval, 1,
        problem.objective_vector, problem.right_hand_side,
        problem.variable_lower_bound, problem.variable_upper_bound,
        problem.num_equalities, device_id, C_NULL))
    end
    Q = problem.objective_matrix
    if length(Q.nzval) > 0
      check(ccall((:pdhg_set_objective_matrix, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Cint),
        h[], length(Q.nzval), Q.colptr, Q.rowval, Q.nzval, 1))
    end
  end
  # zeros(...) state of pdhg.jl:805-819
  state = HipSolverState(h[], n, m, 0.0, 1.0, false, 0.0, 0, nothing, nothing)
  finalizer(s -> ccall((:pdhg_destroy, LIB), Cvoid, (Ptr{Cvoid},), s.handle), state)
  return state
end

"128 opaque bytes naming a new RCCL communicator (rank 0 creates, the host ships them to every rank)."
function unique_id()
  id = zeros(UInt8, UNIQUE_ID_BYTES)
  check(ccall((:pdhg_dist_get_unique_id, LIB), Cint, (Ptr{UInt8},), id))
  return id
end

"""
The library's nnz-balanced contiguous row partition of a global matrix (`pdhg_partition_rows`,
host-only): `world + 1` ascending 0-based bounds; rank r owns rows bounds[r+1]+1 : bounds[r+2]
in Julia's 1-based terms.
"""
function partition_rows(A::SparseMatrixCSC{Float64,Int64}, world::Integer)
  bounds = zeros(Int64, world + 1)
  GC.@preserve A check(ccall((:pdhg_partition_rows, LIB), Cint,
    (Int64, Int64, Ptr{Int64}, Ptr{Int64}, Cint, Cint, Ptr{Int64}),
    size(A, 1), size(A, 2), A.colptr, A.rowval, 1, world, bounds))
  return bounds
end

"""
One process per GPU with RANK-LOCAL ingest (`pdhg_create_dist_rows`): `rows` is this rank's
row block `A[bounds[rank+1]+1 : bounds[rank+2], :]` of the global constraint matrix (so its row
indices are already rebased), `b_rows` the matching right-hand sides; `c, lb, ub` are the
global vectors and `num_equalities` is global.  `dist = (id, rank, world)` as for
`HipSolverState(problem; dist = ...)`.  No rank needs the whole matrix.
"""
function HipSolverStateFromRows(m_global::Integer, bounds::Vector{Int64},
                                rows::SparseMatrixCSC{Float64,Int64}, c::Vector{Float64},
                                b_rows::Vector{Float64}, lb::Vector{Float64}, ub::Vector{Float64},
                                num_equalities::Integer, dist; device_id::Integer = -1)
  id, rank, world = dist
  n = size(rows, 2)
  h = Ref{Ptr{Cvoid}}(C_NULL)
  GC.@preserve rows c b_rows lb ub bounds id check(ccall((:pdhg_create_dist_rows, LIB), Cint,
    (Ref{Ptr{Cvoid}}, Int64, Int64, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Cint,
     Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Cint, Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint),
    h, m_global, n, bounds, length(rows.nzval), rows.colptr, rows.rowval, rows.nzval, 1,
    c, b_rows, lb, ub, num_equalities, device_id, C_NULL, id, rank, world))
  state = HipSolverState(h[], n, m_global, 0.0, 1.0, false, 0.0, 0, nothing, nothing)
  finalizer(s -> ccall((:pdhg_destroy, LIB), Cvoid, (Ptr{Cvoid},), s.handle), state)
  return state
end

"Which RCCL the library bound at run time (`pdhg_rccl_info`); errors if it is unavailable or of another major version."
function rccl_info()
  compiled = Ref{Cint}(0); runtime = Ref{Cint}(0)
  path = zeros(UInt8, 1024)
  check(ccall((:pdhg_rccl_info, LIB), Cint, (Ref{Cint}, Ref{Cint}, Ptr{UInt8}, Cint),
    compiled, runtime, path, length(path)))
  return (compiled_version = compiled[], runtime_version = runtime[],
          path = unsafe_string(pointer(path)))
end

"Checksums of every device array of both layouts (`pdhg_layout_checksums`): equal checksums, bit-identical layouts."
function layout_checksums(s::HipSolverState)
  out = zeros(UInt64, 32)
  check(ccall((:pdhg_layout_checksums, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt64}), s.handle, out))
  return out
end

"HIP-event bracket of empty launches: (ms for one, ms per further launch in the same bracket) (`pdhg_measure_launch_overhead`)."
function measure_launch_overhead(s::HipSolverState, reps::Integer = 20)
  out = zeros(2)
  check(ccall((:pdhg_measure_launch_overhead, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), s.handle, reps, out))
  return out[1], out[2]
end

"Self-test of the check kernels' shared wave reduction: (totals compared, totals whose bits differ: 0 expected) (`pdhg_selftest_wave_sums`)."
function selftest_wave_sums(s::HipSolverState, seed::Integer = 0)
  out = zeros(Int64, 2)
  check(ccall((:pdhg_selftest_wave_sums, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}), s.handle, seed, out))
  return out[1], out[2]
end

"Host-side cost of the trial steps so far: (trials, seconds issuing, seconds waiting) (`pdhg_host_issue_stats`)."
function host_issue_stats(s::HipSolverState)
  trials = Ref{Int64}(0); issue = Ref{Float64}(0.0); wait = Ref{Float64}(0.0)
  check(ccall((:pdhg_host_issue_stats, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Float64}, Ref{Float64}),
    s.handle, trials, issue, wait))
  return trials[], issue[], wait[]
end

function dist_info(s::HipSolverState)
  info = zeros(Int64, 8)
  check(ccall((:pdhg_dist_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), s.handle, info))
  return (world = info[1], local_ranks = info[2], rank = info[3], backend = info[4],
          rows = (info[5], info[6]), columns = (info[7], info[8]))
end

# ------------------------------------------------------------------ hot path

function trial_step(s::HipSolverState, step_size, primal_weight, theta = 1.0)
  out = zeros(5)
  check(ccall((:pdhg_trial_step, LIB), Cint, (Ptr{Cvoid}, Float64, Float64, Float64, Ptr{Float64}),
    s.handle, step_size, primal_weight, theta, out))
  return out
end

trial_primal(s::HipSolverState, step_size, primal_weight) =
  check(ccall((:pdhg_trial_primal, LIB), Cint, (Ptr{Cvoid}, Float64, Float64), s.handle, step_size, primal_weight))

function trial_dual(s::HipSolverState, step_size, primal_weight, theta)
  out = zeros(5)
  check(ccall((:pdhg_trial_dual, LIB), Cint, (Ptr{Cvoid}, Float64, Float64, Float64, Ptr{Float64}),
    s.handle, step_size, primal_weight, theta, out))
  return out
end

accept(s::HipSolverState, w) =
  check(ccall((:pdhg_accept, LIB), Cint, (Ptr{Cvoid}, Float64), s.handle, w))

add_current_primal_to_average(s::HipSolverState, w) =
  check(ccall((:pdhg_add_current_primal_to_average, LIB), Cint, (Ptr{Cvoid}, Float64), s.handle, w))

function average_info(s::HipSolverState)
  counts = zeros(Int64, 2); weights = zeros(2)
  check(ccall((:pdhg_get_average_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Float64}), s.handle, counts, weights))
  return counts[1], counts[2], weights[1], weights[2]
end

"""
The whole adaptive take_step in one ccall (pdhg_take_step_adaptive: the same loop as the
method below, with its scalar part in C).  Julia's ccall overhead is negligible, so the
method below drives trial_step/accept itself; this binding is for hosts that prefer one call.
"""
function take_step_adaptive_native!(s::HipSolverState, reduction_exponent, growth_exponent)
  step = Ref{Float64}(s.step_size); its = Ref{Int64}(s.total_number_iterations)
  kkt = Ref{Float64}(s.cumulative_kkt_passes); err = Ref{Cint}(0)
  check(ccall((:pdhg_take_step_adaptive, LIB), Cint,
    (Ptr{Cvoid}, Float64, Float64, Ref{Float64}, Float64, Ref{Int64}, Ref{Float64}, Ref{Cint}),
    s.handle, reduction_exponent, growth_exponent, step, s.primal_weight, its, kkt, err))
  s.step_size = step[]; s.total_number_iterations = its[]; s.cumulative_kkt_passes = kkt[]
  s.numerical_error = s.numerical_error || err[] != 0
end

"""
`n_steps` adaptive take_steps in one ccall (pdhg_take_steps_adaptive): the iterations
optimize() runs between two termination evaluations.  The library takes them several per kernel
launch with the step rule on the device (bitwise the per-step calls, 15-60 % faster on small and
medium LPs), which is why optimize() below uses it.  Returns the number of take_steps done (fewer
than `n_steps` only after a numerical error).
"""
function take_steps_adaptive_native!(s::HipSolverState, n_steps::Integer, reduction_exponent, growth_exponent)
  step = Ref{Float64}(s.step_size); its = Ref{Int64}(s.total_number_iterations)
  kkt = Ref{Float64}(s.cumulative_kkt_passes); err = Ref{Cint}(0); done = Ref{Int64}(0)
  check(ccall((:pdhg_take_steps_adaptive, LIB), Cint,
    (Ptr{Cvoid}, Int64, Float64, Float64, Ref{Float64}, Float64, Ref{Int64}, Ref{Float64}, Ref{Cint}, Ref{Int64}),
    s.handle, n_steps, reduction_exponent, growth_exponent, step, s.primal_weight, its, kkt, err, done))
  s.step_size = step[]; s.total_number_iterations = its[]; s.cumulative_kkt_passes = kkt[]
  s.numerical_error = s.numerical_error || err[] != 0
  return done[]
end

"take_step(::AdaptiveStepsizeParams, ...) -- pdhg.jl:653-731 with the vector work on the GPU."
function FirstOrderLp.take_step(step_params::FirstOrderLp.AdaptiveStepsizeParams,
                                problem, s::HipSolverState)
  step_size = s.step_size
  done = false
  while !done
    s.total_number_iterations += 1
    raw = trial_step(s, step_size, s.primal_weight)
    # compute_interaction_and_movement (pdhg.jl:527-549) from the raw sums
    interaction = abs(raw[1]) + abs(raw[5])
    movement = 0.5 * s.primal_weight * sqrt(raw[2])^2 + (0.5 / s.primal_weight) * sqrt(raw[3])^2
    s.cumulative_kkt_passes += 1
    if movement == 0.0
      s.numerical_error = true
      break
    end
    step_size_limit = interaction > 0 ? movement / interaction : Inf
    if step_size <= step_size_limit
      accept(s, s.step_size)   # weight = step size on entry (pdhg.jl:512)
      done = true
    end
    k1 = s.total_number_iterations + 1
    first_term = (1 - k1^(-step_params.reduction_exponent)) * step_size_limit
    second_term = (1 + k1^(-step_params.growth_exponent)) * step_size
    step_size = min(first_term, second_term)
  end
  s.step_size = step_size
end

"take_step(::ConstantStepsizeParams, ...) -- pdhg.jl:737-767."
function FirstOrderLp.take_step(step_params::FirstOrderLp.ConstantStepsizeParams,
                                problem, s::HipSolverState)
  trial_step(s, s.step_size, s.primal_weight)
  s.cumulative_kkt_passes += 1
  accept(s, s.step_size)
end

"take_step(::MalitskyPockStepsizeParameters, ...) -- pdhg.jl:555-647."
function FirstOrderLp.take_step(step_params::FirstOrderLp.MalitskyPockStepsizeParameters,
                                problem, s::HipSolverState)
  if !FirstOrderLp.is_linear_programming_problem(problem)
    error("Malitsky and Pock linesearch is only supported for linear programming problems.")
  end
  step_size = s.step_size
  ratio_step_sizes = s.ratio_step_sizes
  done = false
  iter = 0
  trial_primal(s, step_size, s.primal_weight)
  s.cumulative_kkt_passes += 0.5
  step_size = step_size +
    step_params.interpolation_coefficient * (sqrt(1 + ratio_step_sizes) - 1) * step_size
  max_iter = 60
  while !done && iter < max_iter
    iter += 1
    s.total_number_iterations += 1
    ratio_step_sizes = step_size / s.step_size
    raw = trial_dual(s, step_size, s.primal_weight, ratio_step_sizes)
    s.cumulative_kkt_passes += 0.5
    norm_delta_dual_product = sqrt(raw[4])
    norm_delta_dual = sqrt(raw[3])
    if step_size * norm_delta_dual_product <= step_params.breaking_factor * norm_delta_dual
      if average_info(s)[1] == 0
        add_current_primal_to_average(s, step_size * ratio_step_sizes)   # pdhg.jl:621-627
      end
      accept(s, s.step_size)
      done = true
    else
      step_size *= step_params.downscaling_factor
    end
  end
  if iter == max_iter && !done
    s.numerical_error = true
    return
  end
  s.step_size = step_size
  s.ratio_step_sizes = ratio_step_sizes
end

# ------------------------------------------------------------------ vector I/O (evaluation cadence)

function get_average(s::HipSolverState)
  x = zeros(s.primal_size); y = zeros(s.dual_size)
  check(ccall((:pdhg_get_average, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), s.handle, x, y))
  return x, y
end

function get_current(s::HipSolverState)
  x = zeros(s.primal_size); y = zeros(s.dual_size)
  check(ccall((:pdhg_get_current, LIB), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), s.handle, x, y, C_NULL))
  return x, y
end

function get_dual_product(s::HipSolverState)
  aty = zeros(s.primal_size)
  check(ccall((:pdhg_get_current, LIB), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), s.handle, C_NULL, C_NULL, aty))
  return aty
end

function get_trial(s::HipSolverState)
  x = zeros(s.primal_size); y = zeros(s.dual_size); aty = zeros(s.primal_size)
  check(ccall((:pdhg_get_trial, LIB), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), s.handle, x, y, aty))
  return x, y, aty
end

set_current(s::HipSolverState, x::Vector{Float64}, y::Vector{Float64}) =
  check(ccall((:pdhg_set_current, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), s.handle, x, y))

function get_point(s::HipSolverState, point)
  x = zeros(s.primal_size); y = zeros(s.dual_size)
  check(ccall((:pdhg_get_point, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}), s.handle, point, x, y))
  return x, y
end

"A*x and A'*y of the (rescaled) device matrix on host vectors."
function spmv(s::HipSolverState, x::Vector{Float64})
  out = zeros(s.dual_size)
  check(ccall((:pdhg_spmv, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), s.handle, x, out))
  return out
end
function spmv_t(s::HipSolverState, y::Vector{Float64})
  out = zeros(s.primal_size)
  check(ccall((:pdhg_spmv_t, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), s.handle, y, out))
  return out
end

reset_average(s::HipSolverState) =
  check(ccall((:pdhg_reset_average, LIB), Cint, (Ptr{Cvoid},), s.handle))
"current .= avg (saddle_point.jl:808-809) and the A'y recompute of pdhg.jl:1018-1022."
restart_to_average(s::HipSolverState) =
  check(ccall((:pdhg_restart_to_average, LIB), Cint, (Ptr{Cvoid},), s.handle))
save_restart_point(s::HipSolverState) =
  check(ccall((:pdhg_save_restart_point, LIB), Cint, (Ptr{Cvoid},), s.handle))

# ------------------------------------------------------------------ rescaling on the device

"rescale_problem (preprocess.jl:631-687) in place on the device; returns (constraint_rescaling, variable_rescaling)."
function rescale!(s::HipSolverState, l_inf_ruiz_iterations, l2_norm_rescaling, pock_chambolle_alpha)
  e = zeros(s.dual_size); d = zeros(s.primal_size)
  use_pc = pock_chambolle_alpha !== nothing
  check(ccall((:pdhg_rescale, LIB), Cint,
    (Ptr{Cvoid}, Cint, Cint, Cint, Float64, Ptr{Float64}, Ptr{Float64}),
    s.handle, l_inf_ruiz_iterations, l2_norm_rescaling ? 1 : 0, use_pc ? 1 : 0,
    use_pc ? Float64(pock_chambolle_alpha) : 0.0, e, d))
  return e, d
end

function get_problem_vectors(s::HipSolverState)
  c = zeros(s.primal_size); b = zeros(s.dual_size)
  lb = zeros(s.primal_size); ub = zeros(s.primal_size)
  check(ccall((:pdhg_get_problem_vectors, LIB), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), s.handle, c, b, lb, ub))
  return c, b, lb, ub
end

"norm(constraint_matrix, Inf) of the resident (rescaled) matrix (pdhg.jl:823-826)."
function matrix_max_abs(s::HipSolverState)
  out = zeros(1)
  check(ccall((:pdhg_matrix_max_abs, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), s.handle, out))
  return out[1]
end

# ------------------------------------------------------------------ evaluation branch on the device

function set_original_problem(s::HipSolverState, scaled_problem::FirstOrderLp.ScaledQpProblem)
  o = scaled_problem.original_qp
  check(ccall((:pdhg_set_original_problem, LIB), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
    s.handle, scaled_problem.constraint_rescaling, scaled_problem.variable_rescaling,
    o.objective_vector, o.right_hand_side, o.variable_lower_bound, o.variable_upper_bound))
end

function eval_point(s::HipSolverState, point)
  out = zeros(24)
  check(ccall((:pdhg_eval_point, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), s.handle, point, out))
  return out
end

function distance_sq_to_restart(s::HipSolverState, point)
  out = zeros(2)
  check(ccall((:pdhg_distance_to_restart, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), s.handle, point, out))
  return out[1], out[2]
end

function point_sumsq(s::HipSolverState, point)
  out = zeros(2)
  check(ccall((:pdhg_point_sumsq, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), s.handle, point, out))
  return out[1], out[2]
end

function trust_region_bound(s::HipSolverState, point, primal_w, dual_w, radius, range, approximate)
  out = zeros(8)
  check(ccall((:pdhg_trust_region_bound, LIB), Cint,
    (Ptr{Cvoid}, Cint, Float64, Float64, Float64, Cint, Cint, Ptr{Float64}),
    s.handle, point, primal_w, dual_w, radius, range, approximate ? 1 : 0, out))
  return out
end

"""
Up to three such problems in one call (abi 10): on medium single handles their searches share one persistent launch;
row p of the result is what `trust_region_bound` returns for problem p, bit for bit.
"""
function trust_region_bounds(s::HipSolverState, points::Vector{Cint}, primal_w, dual_w, radii::Vector{Float64},
                             ranges::Vector{Cint}, approximate)
  count = length(points)
  out = zeros(8 * count)
  check(ccall((:pdhg_trust_region_bounds, LIB), Cint,
    (Ptr{Cvoid}, Cint, Ptr{Cint}, Float64, Float64, Ptr{Float64}, Ptr{Cint}, Cint, Ptr{Float64}),
    s.handle, count, points, primal_w, dual_w, radii, ranges, approximate ? 1 : 0, out))
  return reshape(out, 8, count)
end

"""
bound_optimal_objective (trust_region_utils.jl:271-360) at a device point, uniform
norm weights per block (define_norms, pdhg.jl:265-277).  The minimising point itself
stays on the device, hence the empty vectors in the result.
"""
function bound(s::HipSolverState, objective_constant, point, primal_w, dual_w, radius, norm, approximate = false)
  if norm == FirstOrderLp.EUCLIDEAN_NORM
    o = trust_region_bound(s, point, primal_w, dual_w, radius, 0, approximate)
    lag = o[1] + objective_constant
    return FirstOrderLp.OptimalObjectiveBoundResult(lag, lag + o[2], lag - o[3], Float64[], Float64[])
  end
  op = trust_region_bound(s, point, primal_w, dual_w, radius, 1, approximate)
  od = trust_region_bound(s, point, primal_w, dual_w, radius, 2, approximate)
  lag = op[1] + objective_constant
  return FirstOrderLp.OptimalObjectiveBoundResult(lag, lag + op[2], lag - od[3], Float64[], Float64[])
end

"""
Several `bound_optimal_objective` problems of one check at once: `requests = [(point, radius), ...]` gives one
OptimalObjectiveBoundResult each -- the very numbers `bound` returns one by one.  Euclidean norm: up to three problems per
`pdhg_trust_region_bounds` call (the average, the current iterate and the last restart point of a restart check,
saddle_point.jl:432-496, 551-596: on medium single handles their searches share one persistent launch); MAX_NORM: the
primal and the dual half of one point share a call.
"""
function bounds(s::HipSolverState, objective_constant, requests, primal_w, dual_w, norm, approximate = false)
  out = FirstOrderLp.OptimalObjectiveBoundResult[]
  if norm == FirstOrderLp.EUCLIDEAN_NORM
    for i in 1:3:length(requests)
      chunk = requests[i:min(i + 2, length(requests))]
      rows = trust_region_bounds(s, Cint[p for (p, _) in chunk], primal_w, dual_w, Float64[r for (_, r) in chunk],
                                 zeros(Cint, length(chunk)), approximate)
      for k in 1:length(chunk)
        lag = rows[1, k] + objective_constant
        push!(out, FirstOrderLp.OptimalObjectiveBoundResult(lag, lag + rows[2, k], lag - rows[3, k], Float64[], Float64[]))
      end
    end
    return out
  end
  for (point, radius) in requests
    rows = trust_region_bounds(s, Cint[point, point], primal_w, dual_w, Float64[radius, radius], Cint[1, 2], approximate)
    lag = rows[1, 1] + objective_constant
    push!(out, FirstOrderLp.OptimalObjectiveBoundResult(lag, lag + rows[2, 1], lag - rows[3, 2], Float64[], Float64[]))
  end
  return out
end

"""
compute_iteration_stats (iteration_stats_utils.jl:356-407) assembled from the raw
sums / maxes of pdhg_eval_point on the UNSCALED point (evaluate_unscaled_iteration_stats,
:413-451).  Index map: include/pdhg_hip.h, pdhg_eval_point.
"""
function iteration_stats(s::HipSolverState, qp_cache, objective_constant, point, termination_criteria,
                         iteration, cumulative_time, cumulative_kkt_passes, step_size, primal_weight,
                         candidate_type)
  r = eval_point(s, point)
  S0, S1, S2, S3, M0, M1, M2, M3 = r[1:8]
  T0, T1, T2, T3, T4, T5 = r[9:14]
  N0, N1, N2, N3, N4, N5 = r[15:20]
  xqx, max_qx = r[21], r[22]
  eps_ratio = termination_criteria.eps_optimal_absolute / termination_criteria.eps_optimal_relative
  ci = FirstOrderLp.ConvergenceInformation()
  ci.primal_objective = objective_constant + T3 + 0.5 * xqx                 # iteration_stats_utils.jl:66-75
  ci.l_inf_primal_residual = max(M0, N2)
  ci.l2_primal_residual = sqrt(S0 + T4)
  ci.relative_l_inf_primal_residual =
    ci.l_inf_primal_residual / (eps_ratio + qp_cache.l_inf_norm_primal_right_hand_side)
  ci.relative_l2_primal_residual =
    ci.l2_primal_residual / (eps_ratio + qp_cache.l2_norm_primal_right_hand_side)
  ci.l_inf_primal_variable = N1
  ci.l2_primal_variable = sqrt(T2)
  ci.dual_objective = (S2 + objective_constant - 0.5 * xqx) + T1            # :180-196
  ci.l_inf_dual_residual = max(M3, N0)
  ci.l2_dual_residual = sqrt(S3 + T0)
  ci.relative_l_inf_dual_residual =
    ci.l_inf_dual_residual / (eps_ratio + qp_cache.l_inf_norm_primal_linear_objective)
  ci.relative_l2_dual_residual =
    ci.l2_dual_residual / (eps_ratio + qp_cache.l2_norm_primal_linear_objective)
  ci.l_inf_dual_variable = M2
  ci.l2_dual_variable = sqrt(S1)
  ci.corrected_dual_objective = ci.l_inf_dual_residual == 0.0 ? ci.dual_objective : -Inf
  gap = abs(ci.primal_objective - ci.dual_objective)
  abs_obj = abs(ci.primal_objective) + abs(ci.dual_objective)
  ci.relative_optimality_gap = gap / (eps_ratio + abs_obj)
  ci.candidate_type = candidate_type

  ii = FirstOrderLp.InfeasibilityInformation()
  sc = N1 != 0.0 ? N1 : 1.0                      # primal ray scaled to unit inf-norm (:296-300)
  ii.max_primal_ray_infeasibility = max(M1, N5) / sc
  ii.primal_ray_linear_objective = T3 / sc
  ii.primal_ray_quadratic_norm = max_qx / sc     # :316-317
  scaling_factor = max(M2, N4)
  if scaling_factor != 0.0
    ii.max_dual_ray_infeasibility = max(M3, N3) / scaling_factor
    ii.dual_ray_objective = (S2 + T5) / scaling_factor
  end
  ii.candidate_type = candidate_type

  stats = FirstOrderLp.IterationStats()
  stats.iteration_number = iteration - 1
  stats.cumulative_kkt_matrix_passes = cumulative_kkt_passes
  stats.cumulative_time_sec = cumulative_time
  stats.convergence_information = [ci]
  stats.infeasibility_information = [ii]
  stats.step_size = step_size
  stats.primal_weight = primal_weight
  stats.method_specific_stats = Dict{String,Float64}()
  return stats
end

# ------------------------------------------------------------------ restart scheme (saddle_point.jl:432-927)

"RestartInfo of saddle_point.jl:158-198 without the vectors: the restart point lives on the device."
mutable struct HipRestartInfo
  last_restart_localized_duality_gap::Union{Nothing,FirstOrderLp.OptimalObjectiveBoundResult}
  last_restart_length::Int64
  primal_distance_moved_last_restart_period::Float64
  dual_distance_moved_last_restart_period::Float64
  gap_reduction_ratio_last_trial::Float64
end
create_last_restart_info() = HipRestartInfo(nothing, 1, 0.0, 0.0, 1.0)    # :200-213

"run_restart_scheme (saddle_point.jl:688-846) with the vector work behind the handle."
function run_restart_scheme(s::HipSolverState, objective_constant, lri::HipRestartInfo, iterations_completed,
                            primal_w, dual_w, primal_weight, verbosity, rp::FirstOrderLp.RestartParameters)
  count_x, count_y, _, _ = average_info(s)
  if !(count_x > 0 && count_y > 0)
    return FirstOrderLp.RESTART_CHOICE_NO_RESTART
  end
  restart_length = count_x
  artificial_restart = false
  do_restart = false
  if restart_length >= rp.artificial_restart_threshold * iterations_completed
    do_restart = true
    artificial_restart = true
  end
  approx = rp.use_approximate_localized_duality_gap
  average_distance_sq = nothing
  candidate_localized_gap = nothing
  candidate_distance_traveled = nothing
  gap_at_last_restart = nothing
  reset_to_average = false
  if rp.restart_scheme != FirstOrderLp.NO_RESTARTS
    # compute_localized_duality_gaps (:432-496): weighted_norm(v, w)^2 == w * sum(v.^2).  The bounds at the average and at
    # the current iterate -- and, for the adaptive-normalized test below (:551-596), the one at the last restart point --
    # are ONE request (pdhg_trust_region_bounds): the same numbers as three `bound` calls, one persistent launch.
    dx2, dy2 = distance_sq_to_restart(s, POINT_AVERAGE)
    average_distance_sq = (dx2, dy2)
    distance_traveled_by_average = sqrt(primal_w * dx2 + dual_w * dy2)
    cx2, cy2 = distance_sq_to_restart(s, POINT_CURRENT)
    distance_traveled_by_current = sqrt(primal_w * cx2 + dual_w * cy2)
    requests = Tuple{Cint,Float64}[(POINT_AVERAGE, distance_traveled_by_average), (POINT_CURRENT, distance_traveled_by_current)]
    if !do_restart && rp.restart_scheme == FirstOrderLp.ADAPTIVE_NORMALIZED
      distance_traveled_last_restart = sqrt(
        lri.primal_distance_moved_last_restart_period^2 * primal_weight +
        lri.dual_distance_moved_last_restart_period^2 / primal_weight)
      push!(requests, (POINT_RESTART, distance_traveled_last_restart))
    end
    gaps = bounds(s, objective_constant, requests, primal_w, dual_w, FirstOrderLp.EUCLIDEAN_NORM, approx)
    gap_at_average, gap_at_current = gaps[1], gaps[2]
    gap_at_last_restart = length(gaps) >= 3 ? gaps[3] : nothing
    reset_to_average = FirstOrderLp.should_reset_to_average(
      gap_at_current, distance_traveled_by_current, gap_at_average, distance_traveled_by_average,
      rp.restart_to_current_metric)
    if reset_to_average
      candidate_localized_gap = gap_at_average
      candidate_distance_traveled = distance_traveled_by_average
    else
      candidate_localized_gap = gap_at_current
      candidate_distance_traveled = distance_traveled_by_current
    end
  end

  if !do_restart
    scheme = rp.restart_scheme
    if scheme == FirstOrderLp.ADAPTIVE_NORMALIZED
      # should_do_adaptive_restart_normalized_duality_gap (:549-596)
      distance_traveled_last_restart = sqrt(
        lri.primal_distance_moved_last_restart_period^2 * primal_weight +
        lri.dual_distance_moved_last_restart_period^2 / primal_weight)
      last_restart = gap_at_last_restart        # (requested with the two candidates above)
      normalized_candidate_gap = FirstOrderLp.get_gap(candidate_localized_gap) / candidate_distance_traveled
      normalized_last_restart_gap = FirstOrderLp.get_gap(last_restart) / distance_traveled_last_restart
      gap_reduction_ratio = normalized_candidate_gap / normalized_last_restart_gap
      if gap_reduction_ratio < rp.necessary_reduction_for_restart
        if gap_reduction_ratio < rp.sufficient_reduction_for_restart
          do_restart = true
        elseif gap_reduction_ratio > lri.gap_reduction_ratio_last_trial
          do_restart = true
        end
      end
      lri.gap_reduction_ratio_last_trial = gap_reduction_ratio
    elseif (scheme == FirstOrderLp.ADAPTIVE_LOCALIZED || scheme == FirstOrderLp.ADAPTIVE_DISTANCE) &&
           lri.last_restart_localized_duality_gap === nothing
      do_restart = true
    elseif scheme == FirstOrderLp.ADAPTIVE_LOCALIZED                       # :597-621
      new_potential = FirstOrderLp.get_gap(candidate_localized_gap) / restart_length
      old_potential = FirstOrderLp.get_gap(lri.last_restart_localized_duality_gap) / lri.last_restart_length
      do_restart = new_potential / old_potential < rp.necessary_reduction_for_restart
    elseif scheme == FirstOrderLp.ADAPTIVE_DISTANCE                        # :623-653
      distance_traveled_last_restart = sqrt(
        lri.primal_distance_moved_last_restart_period^2 * primal_weight +
        lri.dual_distance_moved_last_restart_period^2 / primal_weight)
      new_potential = candidate_distance_traveled / restart_length
      old_potential = distance_traveled_last_restart / lri.last_restart_length
      do_restart = new_potential / old_potential < rp.necessary_reduction_for_restart
    elseif scheme == FirstOrderLp.FIXED_FREQUENCY && rp.restart_frequency_if_fixed <= restart_length
      do_restart = true
    end
  end
  if !do_restart
    return FirstOrderLp.RESTART_CHOICE_NO_RESTART
  end
  if verbosity >= 4
    print(reset_to_average ? "  Restarted to average" : "  Restarted to current")
    println(" after ", rpad(restart_length, 4), " iterations", artificial_restart ? "*" : "")
  end
  # update_last_restart_info (:893-927): distances of the average from the OLD restart point
  if average_distance_sq === nothing
    average_distance_sq = distance_sq_to_restart(s, POINT_AVERAGE)
  end
  lri.primal_distance_moved_last_restart_period = sqrt(primal_w * average_distance_sq[1]) / sqrt(primal_weight)
  lri.dual_distance_moved_last_restart_period = sqrt(dual_w * average_distance_sq[2]) * sqrt(primal_weight)
  lri.last_restart_length = restart_length
  lri.last_restart_localized_duality_gap = candidate_localized_gap
  # current .= avg (if chosen) ; reset_solution_weighted_average ; restart point .= current
  if reset_to_average
    restart_to_average(s)
  end
  reset_average(s)
  save_restart_point(s)
  return reset_to_average ? FirstOrderLp.RESTART_CHOICE_RESTART_TO_AVERAGE :
         FirstOrderLp.RESTART_CHOICE_WEIGHTED_AVERAGE_RESET
end

"compute_new_primal_weight (saddle_point.jl:862-891)."
function compute_new_primal_weight(lri::HipRestartInfo, primal_weight, smoothing, verbosity)
  primal_distance = lri.primal_distance_moved_last_restart_period
  dual_distance = lri.dual_distance_moved_last_restart_period
  if primal_distance > eps() && dual_distance > eps()
    new_primal_weight_estimate = dual_distance / primal_distance
    log_primal_weight = smoothing * log(new_primal_weight_estimate) + (1 - smoothing) * log(primal_weight)
    primal_weight = exp(log_primal_weight)
    if verbosity >= 4
      println("  New computed primal weight is ", primal_weight)
    end
  end
  return primal_weight
end

"update_objective_bound_estimates (saddle_point.jl:1015-1047), uniform norm weights."
function update_objective_bound_estimates(method_specific_stats, s::HipSolverState, objective_constant,
                                          point, primal_w, dual_w)
  sx2, sy2 = point_sumsq(s, point)
  estimated_primal_distance_to_optimality = max(1e-8, sqrt(primal_w * sx2))
  estimated_dual_distance_to_optimality = max(1e-8, sqrt(dual_w * sy2))
  # (through `bounds`: the primal and the dual half of MAX_NORM are two trust-region problems, one launch on the device)
  gap = bounds(s, objective_constant, Tuple{Cint,Float64}[(Cint(point), 1.0)],
               primal_w / estimated_primal_distance_to_optimality^2,
               dual_w / estimated_dual_distance_to_optimality^2, FirstOrderLp.MAX_NORM, false)[1]
  method_specific_stats["lagrangian_value"] = gap.lagrangian_value
  method_specific_stats["estimated_lower_bound"] = gap.lower_bound_value
  method_specific_stats["estimated_upper_bound"] = gap.upper_bound_value
end

"estimate_maximum_singular_value (pdhg.jl:414-440) with the two products on the device."
function estimate_maximum_singular_value(s::HipSolverState; probability_of_failure = 0.01,
                                         desired_relative_error = 0.1, seed::Int64 = 1)
  epsilon = 1.0 - (1.0 - desired_relative_error)^2
  x = randn(Random.MersenneTwister(seed), s.primal_size)
  number_of_power_iterations = 0
  while FirstOrderLp.power_method_failure_probability(s.primal_size, epsilon, number_of_power_iterations) >
        probability_of_failure
    x = x / norm(x, 2)
    x = spmv_t(s, spmv(s, x))
    number_of_power_iterations += 1
  end
  return sqrt(dot(x, spmv_t(s, spmv(s, x))) / norm(x, 2)^2), number_of_power_iterations
end

# ------------------------------------------------------------------ optimize

"PdhgParameters plus where to run: `devices === nothing` = the current GPU, otherwise the GPUs to row-partition over."
struct HipPdhgParameters
  pdhg::FirstOrderLp.PdhgParameters
  devices::Union{Nothing,Vector{Int}}
end
HipPdhgParameters(p::FirstOrderLp.PdhgParameters; devices = nothing) =
  HipPdhgParameters(p, devices === nothing ? nothing : collect(Int, devices))

"""
optimize(params, original_problem) -- src/primal_dual_hybrid_gradient.jl:782-1049,
vector work on the device.  Differences from the reference, line by line, are listed
in INTEGRATION.md section 2.
"""
function FirstOrderLp.optimize(hp::HipPdhgParameters,
                               original_problem::FirstOrderLp.QuadraticProgrammingProblem)
  params = hp.pdhg
  FirstOrderLp.validate(original_problem)
  qp_cache = FirstOrderLp.cached_quadratic_program_info(original_problem)
  if params.primal_importance <= 0 || !isfinite(params.primal_importance)
    error("primal_importance must be positive and finite")
  end

  # :788-795  rescale_problem -> upload the ORIGINAL problem, rescale on the device
  solver_state = HipSolverState(original_problem; devices = hp.devices)
  constraint_rescaling, variable_rescaling = rescale!(
    solver_state, params.l_inf_ruiz_iterations, params.l2_norm_rescaling, params.pock_chambolle_alpha)
  c_s, b_s, lb_s, ub_s = get_problem_vectors(solver_state)
  primal_size = solver_state.primal_size
  dual_size = solver_state.dual_size
  # host copy of the scaled problem: vectors only (the scaled matrices live on the device)
  problem = FirstOrderLp.QuadraticProgrammingProblem(
    lb_s, ub_s, spzeros(primal_size, primal_size), c_s, original_problem.objective_constant,
    spzeros(dual_size, primal_size), b_s, original_problem.num_equalities)
  scaled_problem = FirstOrderLp.ScaledQpProblem(original_problem, problem, constraint_rescaling, variable_rescaling)
  set_original_problem(solver_state, scaled_problem)
  objective_constant = original_problem.objective_constant

  # :821-841  initial step size
  matrix_inf_norm = matrix_max_abs(solver_state)
  if params.step_size_policy_params isa FirstOrderLp.AdaptiveStepsizeParams
    solver_state.cumulative_kkt_passes += 0.5
    solver_state.step_size = 1.0 / matrix_inf_norm
  elseif params.step_size_policy_params isa FirstOrderLp.MalitskyPockStepsizeParameters
    solver_state.cumulative_kkt_passes += 0.5
    solver_state.step_size = 1.0 / matrix_inf_norm
    solver_state.ratio_step_sizes = 1.0
  else
    desired_relative_error = 0.2
    maximum_singular_value, number_of_power_iterations = estimate_maximum_singular_value(
      solver_state, probability_of_failure = 0.001, desired_relative_error = desired_relative_error)
    solver_state.step_size = (1 - desired_relative_error) / maximum_singular_value
    solver_state.cumulative_kkt_passes += number_of_power_iterations
  end

  KKT_PASSES_PER_TERMINATION_EVALUATION = 2.0

  if params.scale_invariant_initial_primal_weight
    solver_state.primal_weight = FirstOrderLp.select_initial_primal_weight(
      problem, ones(primal_size), ones(dual_size), params.primal_importance, params.verbosity)
  else
    solver_state.primal_weight = params.primal_importance
  end

  primal_weight_update_smoothing = params.restart_params.primal_weight_update_smoothing
  iteration_stats_log = FirstOrderLp.IterationStats[]
  start_time = time()
  time_spent_doing_basic_algorithm = 0.0
  last_restart_info = create_last_restart_info()     # the restart point (zeros) is the device's initial one

  termination_criteria = params.termination_criteria
  iteration_limit = termination_criteria.iteration_limit
  termination_evaluation_frequency = params.termination_evaluation_frequency
  solver_state.numerical_error = false
  FirstOrderLp.display_iteration_stats_heading(params.verbosity)

  iteration = 0
  while true
    iteration += 1
    if mod(iteration - 1, termination_evaluation_frequency) == 0 ||
       iteration == iteration_limit + 1 ||
       iteration <= 10 ||
       solver_state.numerical_error
      solver_state.cumulative_kkt_passes += KKT_PASSES_PER_TERMINATION_EVALUATION
      # :902-910  which point is evaluated
      count_x, count_y, _, _ = average_info(solver_state)
      avg_point = (solver_state.numerical_error || count_x == 0 || count_y == 0) ? POINT_CURRENT : POINT_AVERAGE

      # :912-927  evaluate_unscaled_iteration_stats -> pdhg_eval_point
      current_iteration_stats = iteration_stats(
        solver_state, qp_cache, objective_constant, avg_point, termination_criteria, iteration,
        time() - start_time, solver_state.cumulative_kkt_passes, solver_state.step_size,
        solver_state.primal_weight, FirstOrderLp.POINT_TYPE_AVERAGE_ITERATE)
      method_specific_stats = current_iteration_stats.method_specific_stats
      method_specific_stats["time_spent_doing_basic_algorithm"] = time_spent_doing_basic_algorithm

      # :932-937  define_norms: uniform weights, kept as two scalars
      primal_weight_norm = 1 / solver_state.step_size * solver_state.primal_weight
      dual_weight_norm = 1 / solver_state.step_size / solver_state.primal_weight
      termination_reason = FirstOrderLp.check_termination_criteria(
        termination_criteria, qp_cache, current_iteration_stats)
      if solver_state.numerical_error && termination_reason == false
        termination_reason = FirstOrderLp.TERMINATION_REASON_NUMERICAL_ERROR
      end
      # :938-945  -> pdhg_point_sumsq + pdhg_trust_region_bound.  The three entries it fills are only ever read from KEPT
      # stats (the solve log; the final log, saddle_point.jl:961-993): a check whose stats are dropped skips them.
      if params.record_iteration_stats || termination_reason != false
        update_objective_bound_estimates(method_specific_stats, solver_state, problem.objective_constant,
                                         avg_point, primal_weight_norm, dual_weight_norm)
      end
      if params.record_iteration_stats || termination_reason != false
        push!(iteration_stats_log, current_iteration_stats)
      end
      if FirstOrderLp.print_to_screen_this_iteration(
        termination_reason, iteration, params.verbosity, termination_evaluation_frequency)
        FirstOrderLp.display_iteration_stats(current_iteration_stats, params.verbosity)
      end

      if termination_reason != false
        # ** Terminate the algorithm ** (the only exit, :973-992); the solution leaves the device here
        avg_primal_solution, avg_dual_solution = get_point(solver_state, avg_point)
        return FirstOrderLp.unscaled_saddle_point_output(
          scaled_problem, avg_primal_solution, avg_dual_solution, termination_reason,
          iteration - 1, iteration_stats_log)
      end

      # :994-1006  -> pdhg_distance_to_restart / pdhg_trust_region_bound / pdhg_restart_to_average ...
      current_iteration_stats.restart_used = run_restart_scheme(
        solver_state, problem.objective_constant, last_restart_info, iteration - 1,
        primal_weight_norm, dual_weight_norm, solver_state.primal_weight, params.verbosity,
        params.restart_params)
      if current_iteration_stats.restart_used != FirstOrderLp.RESTART_CHOICE_NO_RESTART
        solver_state.primal_weight = compute_new_primal_weight(
          last_restart_info, solver_state.primal_weight, primal_weight_update_smoothing, params.verbosity)
        solver_state.ratio_step_sizes = 1.0
      end
      # :1017-1022  A'y after RESTART_TO_AVERAGE was recomputed inside pdhg_restart_to_average
    end

    time_spent_doing_basic_algorithm_checkpoint = time()
    if params.step_size_policy_params isa FirstOrderLp.AdaptiveStepsizeParams
      # This iteration's take_step and those of the iterations up to (not including) the next one the test above fires
      # on -- the reference does nothing else on them (pdhg.jl:862-1046) -- in ONE library call: the library takes them
      # several per kernel launch with the step rule on the device (bitwise the per-step calls).
      next_evaluation = (div(iteration - 1, termination_evaluation_frequency) + 1) * termination_evaluation_frequency + 1
      if iteration < 10
        next_evaluation = iteration + 1
      end
      if iteration < iteration_limit + 1
        next_evaluation = min(next_evaluation, iteration_limit + 1)
      end
      iteration += take_steps_adaptive_native!(
        solver_state, next_evaluation - iteration,
        params.step_size_policy_params.reduction_exponent, params.step_size_policy_params.growth_exponent) - 1
    else
      FirstOrderLp.take_step(params.step_size_policy_params, original_problem, solver_state)
    end
    time_spent_doing_basic_algorithm += time() - time_spent_doing_basic_algorithm_checkpoint
  end
end

# ------------------------------------------------------------------ measurement

profile_enable(s::HipSolverState, on::Bool) =
  check(ccall((:pdhg_profile_enable, LIB), Cint, (Ptr{Cvoid}, Cint), s.handle, on ? 1 : 0))

function profile_read(s::HipSolverState, kernel_id)
  launches = Ref{Int64}(0); ms = Ref{Float64}(0.0)
  check(ccall((:pdhg_profile_read, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Int64}, Ref{Float64}),
    s.handle, kernel_id, launches, ms))
  return launches[], ms[]
end

kernel_algorithmic_bytes(s::HipSolverState, kernel_id) =
  ccall((:pdhg_kernel_algorithmic_bytes, LIB), Int64, (Ptr{Cvoid}, Cint), s.handle, kernel_id)
kernel_name(s::HipSolverState, kernel_id) =
  unsafe_string(ccall((:pdhg_kernel_name, LIB), Cstring, (Ptr{Cvoid}, Cint), s.handle, kernel_id))

function layout_info(s::HipSolverState)
  info = zeros(Int64, 16)
  check(ccall((:pdhg_layout_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), s.handle, info))
  return info
end

"The resident layouts and every choice `pdhg_create` made for them, timed ones included, as JSON text (`pdhg_layout_describe`)."
function layout_describe(s::HipSolverState)
  need = ccall((:pdhg_layout_describe, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint), s.handle, C_NULL, Cint(0))
  need >= 0 || check(need)
  buf = zeros(UInt8, need + 1)
  ccall((:pdhg_layout_describe, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint), s.handle, buf, Cint(need + 1))
  return unsafe_string(pointer(buf))
end

function measure_triad(s::HipSolverState, len::Int64 = 1 << 26, reps::Int = 5)
  gbps = Ref{Float64}(0.0)
  check(ccall((:pdhg_measure_triad, LIB), Cint, (Ptr{Cvoid}, Int64, Cint, Ref{Float64}), s.handle, len, reps, gbps))
  return gbps[]
end

# the gather rate the tiled sweep's access pattern reaches with nothing else in the kernel (benchmark aid)
function measure_sweep_ceiling(s::HipSolverState, rows::Int64, cols::Int64, nnz::Int64, reps::Int = 3)
  out = zeros(Float64, 6)
  check(ccall((:pdhg_measure_sweep_ceiling, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Int64, Cint, Ptr{Float64}), s.handle, rows, cols, nnz, reps, out))
  return out
end

# phase timeline of the last one-launch trial (PDHG_COOP_TRACE=1); `nothing` when nothing was traced
function trial_timeline(s::HipSolverState)
  out = zeros(Float64, 14)
  rc = ccall((:pdhg_trial_timeline, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), s.handle, out)
  return rc == 0 ? out : nothing
end

end # module
