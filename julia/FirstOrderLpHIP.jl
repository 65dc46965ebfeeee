# FirstOrderLpHIP.jl -- thin `ccall` shim over libpdhg_hip.so (include/pdhg_hip.h).
#
# Drop-in for the one hot path of FirstOrderLp.jl: a device-backed solver state
# and `take_step` methods that `FirstOrderLp.optimize` (src/primal_dual_hybrid_gradient.jl:782-1049)
# can call at :1044 in place of the CPU ones.  NOT executed in the build
# environment (no Julia there); it mirrors firstorderlp.jl_amd/engine.py and
# primal_dual_hybrid_gradient.py one to one.
module FirstOrderLpHIP

import FirstOrderLp
const LIB = get(ENV, "PDHG_HIP_LIB", "libpdhg_hip.so")

mutable struct HipSolverState
  handle::Ptr{Cvoid}
  step_size::Float64
  primal_weight::Float64
  numerical_error::Bool
  cumulative_kkt_passes::Float64
  total_number_iterations::Int64
  ratio_step_sizes::Float64
end

check(rc::Cint) = rc == 0 || error("pdhg_hip error $rc: " *
  unsafe_string(ccall((:pdhg_last_error, LIB), Cstring, ())))

"Ingest the *rescaled* problem exactly as Julia stores it (CSC, Int64, 1-based)."
function HipSolverState(problem::FirstOrderLp.QuadraticProgrammingProblem)
  A = problem.constraint_matrix
  m, n = size(A)
  h = Ref{Ptr{Cvoid}}(C_NULL)
  GC.@preserve A problem begin
    check(ccall((:pdhg_create, LIB), Cint,
      (Ref{Ptr{Cvoid}}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Cint,
       Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Cint, Ptr{Cvoid}),
      h, m, n, length(A.nzval), A.colptr, A.rowval, A.nzval, 1,
      problem.objective_vector, problem.right_hand_side,
      problem.variable_lower_bound, problem.variable_upper_bound,
      problem.num_equalities, -1, C_NULL))
    Q = problem.objective_matrix
    if length(Q.nzval) > 0
      check(ccall((:pdhg_set_objective_matrix, LIB), Cint,
        (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Cint),
        h[], length(Q.nzval), Q.colptr, Q.rowval, Q.nzval, 1))
    end
  end
  state = HipSolverState(h[], 0.0, 1.0, false, 0.0, 0, 1.0)
  finalizer(s -> ccall((:pdhg_destroy, LIB), Cvoid, (Ptr{Cvoid},), s.handle), state)
  return state
end

function trial_step(s::HipSolverState, step_size, primal_weight, theta = 1.0)
  out = zeros(5)
  check(ccall((:pdhg_trial_step, LIB), Cint, (Ptr{Cvoid}, Float64, Float64, Float64, Ptr{Float64}),
    s.handle, step_size, primal_weight, theta, out))
  return out
end

accept(s::HipSolverState, w) =
  check(ccall((:pdhg_accept, LIB), Cint, (Ptr{Cvoid}, Float64), s.handle, w))

"take_step(::AdaptiveStepsizeParams, ...) -- pdhg.jl:653-731 with the vector work on the GPU."
function FirstOrderLp.take_step(step_params::FirstOrderLp.AdaptiveStepsizeParams,
                                problem, s::HipSolverState)
  step_size = s.step_size
  done = false
  while !done
    s.total_number_iterations += 1
    raw = trial_step(s, step_size, s.primal_weight)
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
    step_size = min((1 - k1^(-step_params.reduction_exponent)) * step_size_limit,
                    (1 + k1^(-step_params.growth_exponent)) * step_size)
  end
  s.step_size = step_size
end

function get_average(s::HipSolverState, n, m)
  x = zeros(n); y = zeros(m)
  check(ccall((:pdhg_get_average, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), s.handle, x, y))
  return x, y
end

function get_current(s::HipSolverState, n, m)
  x = zeros(n); y = zeros(m)
  check(ccall((:pdhg_get_current, LIB), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), s.handle, x, y, C_NULL))
  return x, y
end

restart_to_average(s::HipSolverState) =
  check(ccall((:pdhg_restart_to_average, LIB), Cint, (Ptr{Cvoid},), s.handle))
reset_average(s::HipSolverState) =
  check(ccall((:pdhg_reset_average, LIB), Cint, (Ptr{Cvoid},), s.handle))

end # module
