// common.hpp -- part of the single translation unit pdhg_hip.hip (included there, in order).
// Constants, error plumbing and device utilities shared by every kernel file.
#pragma once

namespace {

constexpr int TPB = 256;             // 4 waves of 64
constexpr int WAVE = 64;
constexpr int BLOCK_NNZ = 2048;      // products staged in LDS per workgroup (16 KiB)
constexpr int UNROLL = BLOCK_NNZ / TPB;
constexpr int MAX_ROWS_PER_BLOCK = 4 * TPB;
constexpr int LONG_CHUNK = BLOCK_NNZ;   // nnz per workgroup for rows longer than BLOCK_NNZ: read like a row block (spmv_kernels.hpp)
constexpr int NUM_XCD = 8;
constexpr int EW_MAX_BLOCKS = 256 * 8;  // elementwise kernels: grid-stride above this
constexpr int FINAL_TPB = 1024;
// tiled-sweep layout (SpMV v2)
constexpr int TW_WPB = 8;              // waves per workgroup (512 threads), 2 workgroups per CU
constexpr int TW_MAX_ROWS = 1272;      // rows owned by one wave: 2 x 8 x 1272 x 8 B fits the 160 KiB LDS
constexpr unsigned TW_PAD = 0xFFFFFFFFu;
#ifndef TWD_U            // dev builds (tools/variants.sh) may override
#define TWD_U 3
#endif
constexpr int TW_U = TWD_U;           // 64-entry chunks prefetched per wave per tile

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
