// common.hpp -- part of the single translation unit pdhg_hip.hip (included there, in order).
// Constants, error plumbing and device utilities shared by every kernel file.
#pragma once

namespace {

constexpr int TPB = 256;             // 4 waves of 64
constexpr int WAVE = 64;
constexpr int BLOCK_NNZ = 2048;      // products staged in LDS per workgroup (16 KiB)
constexpr int UNROLL = BLOCK_NNZ / TPB;
#ifndef PDHG_MAX_ROWS_PER_BLOCK
#define PDHG_MAX_ROWS_PER_BLOCK (4 * TPB)
#endif
constexpr int MAX_ROWS_PER_BLOCK = PDHG_MAX_ROWS_PER_BLOCK;      // (dev: -DPDHG_MAX_ROWS_PER_BLOCK=512 -- fewer row trips per block of very short rows)
// Environment variables.  The documented run-time knobs (include/pdhg_hip.h has the table) are read with getenv as they
// are; everything else -- tuning constants, negative-result paths kept for the measurements that settled them, fault
// injection for tests -- is a DEVELOPMENT variable and is only honoured when PDHG_DEV=1 is set as well (tests/conftest.py
// and the tools set it): a stray PDHG_TW_ROWS in a user's shell cannot change what the library does.
inline const char *dev_env(const char *name) {
  static const bool on = [] { const char *e = getenv("PDHG_DEV"); return e && e[0] == '1'; }();
  return on ? getenv(name) : nullptr;
}
constexpr int LONG_CHUNK = BLOCK_NNZ;   // nnz per workgroup for rows longer than BLOCK_NNZ: read like a row block (spmv_kernels.hpp)
// Rows with more entries than CsrDev::long_thr go to the long-row kernels (chunks of LONG_CHUNK entries + ordered combine)
// instead of a row block / the tiled sweep.  BLOCK_NNZ is the capacity limit and the default; PDHG_LONG_THR (dev knob,
// read when a layout is built and kept with it) lowers it: hub rows of a power-law graph then leave the sweep, whose
// lanes sum a row's entries inside a tile one after the other.
inline int long_row_threshold_from_env() {
  const char *e = dev_env("PDHG_LONG_THR");
  const int v = e ? atoi(e) : BLOCK_NNZ;
  return v < 16 ? 16 : (v > BLOCK_NNZ ? BLOCK_NNZ : v);
}
constexpr int NUM_XCD = 8;
constexpr int EW_MAX_BLOCKS = 256 * 8;  // elementwise kernels: grid-stride above this
constexpr int FINAL_TPB = 1024;
// tiled-sweep layout (SpMV v2)
constexpr int TW_WPB = 8;              // waves per workgroup (512 threads), 2 workgroups per CU
constexpr int TW_MAX_ROWS = 1272;      // rows owned by one wave: 2 x 8 x 1272 x 8 B fits the 160 KiB LDS
constexpr unsigned TW_PAD = 0xFFFFFFFFu;
#ifndef TWD_U            // dev builds (tools/variants.sh) may override
#define TWD_U 4
#endif
// 64-entry chunks prefetched per wave per tile.  4 since round 5 (73 - 89 VGPRs: still four waves per SIMD): a (wave, tile)
// cell beyond the window costs its whole workgroup a second step, and cells of 95 +- 30 entries (rows clustered inside
// a tile) passed 192 often enough for 8 % of the product's time; on uniform matrices 3, 4 and 5 measure the same.
constexpr int TW_U = TWD_U;

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

int fail_hip(hipError_t e, const char *what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return (int)e > 0 ? (int)e : 999;
}

#define HIP_TRY(expr)                                                        \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      g_last_error = std::string(#expr) + ": " + hipGetErrorString(_e);      \
      return (int)_e > 0 ? (int)_e : 999;                                    \
    }                                                                        \
  } while (0)

// ---------------------------------------------------------------- roctx ranges (tracing)
// PDHG_ROCTX=1: the C-ABI entry points and every fused product push / pop a named roctx range, so that a
// `rocprofv3 --marker-trace --kernel-trace` timeline shows which take_step / product / evaluation a kernel belongs to
// (SURVEY section 5 lists tracing among the auxiliary subsystems; the reference itself has Julia's @info logging only).
// The marker library is bound at run time -- librocprofiler-sdk-roctx (what rocprofv3 listens to), else libroctx64 --
// and the ranges are no-ops when the variable is unset or neither library is there: no link-time dependency.
struct RoctxApi {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  bool on = false;
};
inline RoctxApi &roctx_api() {
  static RoctxApi api = [] {
    RoctxApi a;
    const char *ev = getenv("PDHG_ROCTX");
    if (!ev || ev[0] == '0') return a;
    for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      a.push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
      a.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (a.push && a.pop) { a.on = true; break; }
    }
    if (!a.on) fprintf(stderr, "[pdhg_hip] PDHG_ROCTX is set but no roctx library could be bound: ranges are off\n");
    return a;
  }();
  return api;
}
struct RoctxRange {
  bool on;
  explicit RoctxRange(const char *name) : on(roctx_api().on) { if (on) roctx_api().push(name); }
  ~RoctxRange() { if (on) roctx_api().pop(); }
  RoctxRange(const RoctxRange &) = delete;
  RoctxRange &operator=(const RoctxRange &) = delete;
};

// ---------------------------------------------------------------- device utils

// Julia's max/min on Float64 for non-NaN inputs, including signed zeros
// (saddle_point.jl:88-91, :115 use min(ub, max(lb, v)) and max(y, 0.0)).
__device__ __forceinline__ double jl_max(double a, double b) {
  return (a > b) ? a : ((b > a) ? b : (signbit(a) ? b : a));
}
__device__ __forceinline__ double jl_min(double a, double b) {
  return (a < b) ? a : ((b < a) ? b : (signbit(a) ? a : b));
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
  return v;
}

// ---- double-double accumulation of the step-acceptance sums -----------------------------------
// The three sums of a trial (dx.(A'y' - A'y), |dx|^2, |dy|^2; pdhg.jl:527-549) are accumulated as unevaluated
// pairs hi + lo (TwoSum: the rounding error of every addition is kept in lo), per lane, through the wave and
// block trees and the second stage, and rounded to one double at the very end.  The result is the correctly
// rounded exact sum of the (double) terms except with probability ~1e-14 per sum, WHATEVER THE ORDER of the
// additions -- so it does not depend on tile widths, grid sizes or launch paths, and a CPU run that adds the
// same terms the same way (the oracle's exact-sums mode, test infrastructure) gets the same bits: free-running
// trajectories then agree bitwise with the CPU restatement instead of drifting apart through the
// discontinuous step-size rule.  Cost: ~7 flops per row and quantity; not measurable.
struct Acc3 {
  double hi[3];
  double lo[3];
};
__device__ __forceinline__ Acc3 acc3_zero() { return Acc3{{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}}; }
// (hi, lo) += t
__device__ __forceinline__ void dd_add(double &hi, double &lo, double t) {
  const double s = hi + t;
  const double bb = s - hi;
  const double e = (hi - (s - bb)) + (t - bb);
  hi = s;
  lo = lo + e;
}
// (hi, lo) += (bh, bl), renormalised
__device__ __forceinline__ void dd_add_dd(double &hi, double &lo, double bh, double bl) {
  const double s = hi + bh;
  const double bb = s - hi;
  const double e = (hi - (s - bb)) + (bh - bb);
  const double l = (lo + bl) + e;
  const double h2 = s + l;
  lo = l - (h2 - s);
  hi = h2;
}
// The scalar part of take_step(::AdaptiveStepsizeParams) after one trial (pdhg.jl:527-549, 691-729), ONE definition
// for the host loop (pdhg_take_step_adaptive) and for the kernel that takes several steps per launch
// (steps_kernel): the same expression trees, IEEE sqrt / divide on both sides, -ffp-contract=off, so the same bits
// (tests/test_gpu_device_loop.py checks the device's sqrt and divide against the host's on random inputs, and whole
// trajectories).  raw: the trial's five sums with raw[4] = 0.5 dx'Q dx already; pow_red / pow_growth:
// (total_number_iterations + 1)^-reduction_exponent / ^-growth_exponent, computed by the HOST's pow on both paths.
struct StepRule {
  int accept, numerical_error;
  double next_step;
};
__host__ __device__ inline StepRule adaptive_step_rule(const double raw[5], double primal_weight, double step_size,
                                                       double pow_red, double pow_growth) {
  StepRule r;
  const double interaction = fabs(raw[0]) + fabs(raw[4]);
  const double nx = sqrt(raw[1]), ny = sqrt(raw[2]);
  const double movement = 0.5 * primal_weight * (nx * nx) + (0.5 / primal_weight) * (ny * ny);
  r.accept = 0;
  r.numerical_error = 0;
  r.next_step = step_size;
  if (movement == 0.0) {       // the algorithm terminates at the beginning of the next iteration
    r.numerical_error = 1;
    return r;
  }
  const double step_size_limit = interaction > 0 ? movement / interaction : INFINITY;
  if (step_size <= step_size_limit) r.accept = 1;
  const double first_term = (1 - pow_red) * step_size_limit;
  const double second_term = (1 + pow_growth) * step_size;
  // Julia's min (pdhg.jl:729): a NaN operand gives NaN, so the step size the reference would
  // carry after a NaN trial is NaN, not the finite operand (`a < b ? a : b` drops the NaN)
  r.next_step = (first_term != first_term || second_term != second_term)
                    ? NAN : ((first_term < second_term) ? first_term : second_term);
  return r;
}

// A load that cannot be served by a stale line of this CU's L1: agent scope (sc1), served by the L2.  The
// multi-step trial kernel re-reads vectors that OTHER compute units rewrote since this CU last read them, and a
// per-workgroup L1 invalidate costs ~50 ns per workgroup and XCD, serialised (tools/grid_barrier_probe).  Relaxed:
// no wait is attached, the loads pipeline like plain ones.
template <bool COH>
__device__ __forceinline__ double ldc(const double *p) {
  if (COH) return __hip_atomic_load(const_cast<double *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
// a double moved across lanes by DPP (two 32-bit moves); lanes without a source, or outside row_mask, receive 0.0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
// Wave-wide double-double sum; the total ends in LANE 63.  DPP moves instead of ds_bpermute shuffles (the six
// steps of a shuffle tree over two doubles were ~0.8 us per block on the one-launch trial's critical path):
// inclusive sums inside each row of 16 lanes (row_shr 1, 2, 4, 8), then lane 15 of rows 0 / 2 into rows 1 / 3
// (row_bcast:15), then lane 31 into rows 2 and 3 (row_bcast:31).  Adding the 0.0 a lane without a source
// receives is exact.
__device__ __forceinline__ void wave_sum_dd(double &hi, double &lo) {
#define PDHG_DD_STEP(CTRL, MASK) do { const double bh = dpp_move<CTRL, MASK>(hi), bl = dpp_move<CTRL, MASK>(lo); dd_add_dd(hi, lo, bh, bl); } while (0)
  PDHG_DD_STEP(0x111, 0xF);   // row_shr:1
  PDHG_DD_STEP(0x112, 0xF);   // row_shr:2
  PDHG_DD_STEP(0x114, 0xF);   // row_shr:4
  PDHG_DD_STEP(0x118, 0xF);   // row_shr:8
  PDHG_DD_STEP(0x142, 0xA);   // row_bcast:15 into rows 1 and 3
  PDHG_DD_STEP(0x143, 0xC);   // row_bcast:31 into rows 2 and 3
#undef PDHG_DD_STEP
}
// The same tree for one double (sum) and for a max of NON-NEGATIVE values; the total ends in LANE 63.  A shuffle
// tree costs six dependent LDS-crossbar round trips per quantity (~0.35 us measured on the evaluation kernels,
// which reduce up to 30 quantities per workgroup); these are VALU moves.
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_move<0x111, 0xF>(v);
  v += dpp_move<0x112, 0xF>(v);
  v += dpp_move<0x114, 0xF>(v);
  v += dpp_move<0x118, 0xF>(v);
  v += dpp_move<0x142, 0xA>(v);
  v += dpp_move<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ double wave_max_nonneg_dpp(double v) {
  v = fmax(v, dpp_move<0x111, 0xF>(v));
  v = fmax(v, dpp_move<0x112, 0xF>(v));
  v = fmax(v, dpp_move<0x114, 0xF>(v));
  v = fmax(v, dpp_move<0x118, 0xF>(v));
  v = fmax(v, dpp_move<0x142, 0xA>(v));
  v = fmax(v, dpp_move<0x143, 0xC>(v));
  return v;
}
// Deterministic block reduction of NQ double-double accumulators; thread 0 returns the totals in acc.
// `red` is LDS [6][THREADS / WAVE] (hi rows 0..2, lo rows 3..5).
template <int NQ, int THREADS>
__device__ __forceinline__ void block_sum_dd(Acc3 &acc, double (*red)[THREADS / WAVE]) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int wid = threadIdx.x / WAVE;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    double h = acc.hi[q], l = acc.lo[q];
    wave_sum_dd(h, l);
    if (lane == WAVE - 1) { red[q][wid] = h; red[3 + q][wid] = l; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      double h = 0.0, l = 0.0;
#pragma unroll
      for (int w = 0; w < THREADS / WAVE; ++w) dd_add_dd(h, l, red[q][w], red[3 + q][w]);
      acc.hi[q] = h;
      acc.lo[q] = l;
    }
  }
}

// Deterministic block reduction of up to 3 per-thread accumulators; thread 0
// of the block returns the totals in acc[].  `red` is LDS [3][TPB/WAVE].
template <int NQ, int THREADS>
__device__ __forceinline__ void block_sum(double (&acc)[3],
                                          double (*red)[THREADS / WAVE]) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int wid = threadIdx.x / WAVE;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double w = wave_sum(acc[q]);
    if (lane == 0) red[q][wid] = w;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < THREADS / WAVE; ++w) s += red[q][w];
      acc[q] = s;
    }
  }
}

}  // namespace
