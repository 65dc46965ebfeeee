// Dev probe (not part of the product): WHY do the product kernels gather at 128-148 G/s when a plain kernel gathers
// L2-hit doubles at 191-250 G/s?  Round-5 counters on spmv_stream_kernel (banded 10M +-50000, every gather an L2 hit):
// TA busy 94 %, TCP_PENDING_STALL 79 % of the CU-cycles, waves 47 % of their life in s_waitcnt: the CU's vector-memory
// path is saturated at 0.21 gathered lines per clock.  Hypothesis: the TCP's outstanding-miss capacity is the resource,
// a request holds its share for its whole latency, and the 6 matrix-stream lines per 64 entries (HBM latency, ~2 us) cost
// about as much of it as 20-30 gather lines (L2-hit latency, ~0.4 us).  This probe runs the stream kernel's skeleton
// (2048-entry blocks, 8 (col, val) load pairs + 8 gathers per lane) with the stream arriving in different ways:
//   0  no stream at all (indices hashed in registers): the gather ceiling of this skeleton
//   1  stream from HBM, non-temporal vector loads (what the product does)
//   2  stream from a 3 MB buffer read over and over (vector loads that HIT L2): what (1) would reach if the stream were in L2
//   3  (1) + the NEXT block's lines touched by SCALAR loads (one s_load_dword per 128 B): a prefetch into L2 through the
//      scalar cache's own path, no vector-memory slot held while HBM answers
//   4  (3) with one scalar load per 64 B
//   5  (1) with plain (temporal) vector loads
//   6  (1) + a FIFTH wave per workgroup that does nothing but touch the next block's lines with scalar loads (and waits for
//      them itself), one workgroup barrier per block as pacing: the compute waves never wait for HBM
//   7  (6) with one scalar load per 64 B
//   8  stream from a 96 MB buffer read over and over (misses L2, fits the 256 MB Infinity Cache)
//   12 (1) with 16-byte loads: every lane reads 8 CONSECUTIVE entries (2 x dwordx4 of col, 4 x dwordx4 of val): 6 load
//      instructions per lane instead of 16, the same 192 lines per block
//   13 (1) with the SWEEP's stream pattern: every resident wave reads 64-entry chunks from a region of its OWN (gridDim x 4
//      far-apart sequential streams instead of one front that moves through the arrays)
//   14 the SLICED JAGGED layout in miniature: rows of 10 entries, 64 rows per wave stored level-major (entry j of row r at
//      slice_base + j * 64 + r), every lane walks ITS row left to right in registers (no LDS, no barrier), reads y and b,
//      writes y' -- what a stream-class product would cost without the products-through-LDS phase
//   9-11  (1) through buffer loads with other cache-policy bits: 9 sc1 nt, 10 sc0 sc1 nt, 11 sc1 (agent scope: no L1 allocation)
// usage: slot_probe [entries_millions=100] [window_doubles=100000] [wgs_per_cu=8] [slide=1]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#ifndef U_
#define U_ 8
#endif
constexpr int TPB = 256, U = U_, BLK = TPB * U, WAVE = 64;

template <int V>
__global__ __launch_bounds__(TPB + WAVE) void probe(const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ x,
                                             double *__restrict__ out, long nblk, int window, long small_blocks, int remap) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double s = 0.0;
  int pf = 0;
  if ((V == 6 || V == 7) && wave == TPB / WAVE) {
    // the prefetch wave: block b + gridDim.x while the four compute waves work on block b
    constexpr int STEP = (V == 6) ? 32 : 16;
    constexpr int NC = BLK * 4 / (STEP * 4), NV = BLK * 8 / (STEP * 4);
    for (long b = blockIdx.x; b < nblk; b += gridDim.x) {
      const long nb = b + gridDim.x;
      if (nb < nblk) {
        const int *pc = col + nb * BLK;
        const int *pv = reinterpret_cast<const int *>(val + nb * BLK);
#pragma unroll 16
        for (int l = 0; l < NC; ++l) pf ^= pc[l * STEP];
#pragma unroll 16
        for (int l = 0; l < NV; ++l) pf ^= pv[l * STEP];
      }
      __syncthreads();
    }
    if (pf == 0x7fffffff) out[0] = 1.0;
    return;
  }
  const long per_xcd = (nblk + 7) / 8;
  for (long b0 = blockIdx.x; b0 < nblk; b0 += gridDim.x) {
    // remap: XCD x (= blockIdx % 8 under round-robin dispatch) walks the contiguous eighth [x * per_xcd, (x + 1) * per_xcd)
    const long b = remap ? ((b0 & 7) * per_xcd + (b0 >> 3)) : b0;
    if (b >= nblk) continue;
    const long bb = (V == 2) ? (b % small_blocks) : (V == 8 ? (b % (small_blocks * 32)) : b);
    const long base = bb * BLK;
    int c[U];
    double v[U];
    if (V == 3 || V == 4) {
      // the next block of THIS workgroup: 2048 x 4 B of col = 64 lines, 2048 x 8 B of val = 128 lines; a quarter per wave
      const long nb = b + gridDim.x;
      if (nb < nblk) {
        const int *pc = col + nb * BLK;
        const int *pv = reinterpret_cast<const int *>(val + nb * BLK);
        constexpr int STEP = (V == 3) ? 32 : 16;          // ints per touched unit: 128 B / 64 B
        constexpr int NC = (BLK * 4 / (STEP * 4)) / 4, NV = (BLK * 8 / (STEP * 4)) / 4;
#pragma unroll
        for (int l = 0; l < NC; ++l) pf ^= pc[(wave * NC + l) * STEP];
#pragma unroll
        for (int l = 0; l < NV; ++l) pf ^= pv[(wave * NV + l) * STEP];
      }
    }
    if (V == 13) {
      const long total_waves = (long)gridDim.x * (TPB / WAVE);
      const long region = (nblk * BLK / total_waves) / (WAVE * U) * (WAVE * U);
      const long it = (b0 - blockIdx.x) / gridDim.x;
      const long my = ((long)blockIdx.x * (TPB / WAVE) + wave) * region + it * (WAVE * U);
      const bool ok = (it + 1) * (WAVE * U) <= region;
#pragma unroll
      for (int i = 0; i < U; ++i) {
        const long k = ok ? my + i * WAVE + (tid & (WAVE - 1)) : 0;
        c[i] = __builtin_nontemporal_load(col + k);
        v[i] = __builtin_nontemporal_load(val + k);
      }
    } else if (V == 12 && U == 8) {
      typedef int vi4 __attribute__((ext_vector_type(4)));
      typedef double vd2 __attribute__((ext_vector_type(2)));
      const vi4 *pc = reinterpret_cast<const vi4 *>(col + base) + tid * 2;
      const vd2 *pv = reinterpret_cast<const vd2 *>(val + base) + tid * 4;
      const vi4 c0 = __builtin_nontemporal_load(pc), c1 = __builtin_nontemporal_load(pc + 1);
      const vd2 v0 = __builtin_nontemporal_load(pv), v1 = __builtin_nontemporal_load(pv + 1), v2 = __builtin_nontemporal_load(pv + 2),
                    v3 = __builtin_nontemporal_load(pv + 3);
      c[0] = c0.x; c[1] = c0.y; c[2] = c0.z; c[3] = c0.w; c[4] = c1.x; c[5] = c1.y; c[6] = c1.z; c[7] = c1.w;
      v[0] = v0.x; v[1] = v0.y; v[2] = v1.x; v[3] = v1.y; v[4] = v2.x; v[5] = v2.y; v[6] = v3.x; v[7] = v3.y;
    } else if (V == 0) {
#pragma unroll
      for (int i = 0; i < U; ++i) {
        unsigned h = (unsigned)(base + tid + i * TPB) * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        c[i] = (int)(h % (unsigned)window);
        v[i] = 1.0;
      }
    } else {
#pragma unroll
      for (int i = 0; i < U; ++i) {
        const long k = base + tid + i * TPB;
        if (V >= 9) {
          constexpr int AUX = (V == 9) ? (16 | 2) : (V == 10 ? (16 | 2 | 1) : 16);
          __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(col + base), 0, BLK * 4, 0x00020000);
          __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)(val + base), 0, BLK * 8, 0x00020000);
          c[i] = __builtin_amdgcn_raw_buffer_load_b32(rc, (tid + i * TPB) * 4, 0, AUX);
          auto q = __builtin_amdgcn_raw_buffer_load_b64(rv, (tid + i * TPB) * 8, 0, AUX);
          v[i] = __builtin_bit_cast(double, q);
        } else if (V == 5) { c[i] = col[k]; v[i] = val[k]; }
        else { c[i] = __builtin_nontemporal_load(col + k); v[i] = __builtin_nontemporal_load(val + k); }
      }
    }
    double xv[U];
#pragma unroll
    for (int i = 0; i < U; ++i) xv[i] = x[c[i]];
#pragma unroll
    for (int i = 0; i < U; ++i) s += v[i] * xv[i];
    if (V == 6 || V == 7) __syncthreads();
  }
  if (pf == 0x7fffffff) s += 1.0;
  out[(long)blockIdx.x * TPB + tid] = s;
}

// V14: rows of ROWLEN entries; a workgroup owns 4 slices of 64 rows (2560 entries) per iteration
constexpr int ROWLEN = 10;
__global__ __launch_bounds__(TPB) void sjds_probe(const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ x,
                                                  const double *__restrict__ y, const double *__restrict__ bvec, double *__restrict__ ynext,
                                                  double *__restrict__ out, long nslices, int remap) {
  const int tid = threadIdx.x, lane = tid & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long ngroups = nslices / 4, per_xcd = (ngroups + 7) / 8;
  double dy2 = 0.0;
  for (long g0 = blockIdx.x; g0 < ngroups; g0 += gridDim.x) {
    const long g = remap ? ((g0 & 7) * per_xcd + (g0 >> 3)) : g0;
    if (g >= ngroups) continue;
    const long slice = g * 4 + wave;
    const long base = slice * (WAVE * ROWLEN);
    const long r = slice * WAVE + lane;
    int c[ROWLEN];
    double v[ROWLEN], xv[ROWLEN];
    const double yo = __builtin_nontemporal_load(y + r), bb = __builtin_nontemporal_load(bvec + r);
#pragma unroll
    for (int j = 0; j < ROWLEN; ++j) { c[j] = __builtin_nontemporal_load(col + base + j * WAVE + lane); v[j] = __builtin_nontemporal_load(val + base + j * WAVE + lane); }
#pragma unroll
    for (int j = 0; j < ROWLEN; ++j) xv[j] = x[c[j]];
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < ROWLEN; ++j) { const double p = v[j] * xv[j]; acc = acc + p; }
    double yn = yo + 0.5 * (bb - acc);
    yn = yn > 0.0 ? yn : 0.0;
    __builtin_nontemporal_store(yn, ynext + r);
    const double d = yn - yo;
    dy2 += d * d;
  }
  out[(long)blockIdx.x * TPB + tid] = dy2;
}

template <typename F>
float time_it(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char **argv) {
  const long count = (argc > 1 ? atol(argv[1]) : 100) * 1000000L / BLK * BLK;
  const int window = argc > 2 ? atoi(argv[2]) : 100000;
  const int per_cu = argc > 3 ? atoi(argv[3]) : 8;
  const int slide = argc > 4 ? atoi(argv[4]) : 1;
  const int remap = argc > 5 ? atoi(argv[5]) : 0;       // 1: every XCD walks a contiguous eighth of the blocks (the product's block order)       // 0: every block gathers from the SAME window (L2 hit rate independent of the grid)
  const long nblk = count / BLK;
  std::vector<int> h((size_t)count);
  std::mt19937_64 rng(1);
  // like a banded matrix: block b gathers from a window that slides with b
  const long n = 10000000;
  for (long b = 0; b < nblk; ++b) {
    const long lo = slide ? (n - window) * b / nblk : 0;
    for (int k = 0; k < BLK; ++k) h[(size_t)(b * BLK + k)] = (int)(lo + (long)(rng() % (unsigned long)window));
  }
  int *col; double *val, *x, *out;
  CK(hipMalloc(&col, count * 4)); CK(hipMalloc(&val, count * 8)); CK(hipMalloc(&x, n * 8));
  CK(hipMemcpy(col, h.data(), count * 4, hipMemcpyHostToDevice));
  CK(hipMemset(val, 0, count * 8)); CK(hipMemset(x, 0, n * 8));
  const int grid = 256 * per_cu;
  CK(hipMalloc(&out, (long)grid * TPB * 8));
  const long small_blocks = (3L << 20) / (BLK * 12);     // ~3 MB of stream, chip-wide (each XCD's L2 sees an eighth of the workgroups)
  printf("entries %ld, window %d doubles (%.0f KB), grid %d (%d workgroups per CU)\n", count, window, window * 8 / 1024.0, grid, per_cu);
  const char *names[14] = {"0 gathers only (hashed indices)", "1 stream from HBM, nt vector loads (product)", "2 stream hits L2 (3 MB re-read)",
                          "3 (1) + scalar prefetch of the next block, 1 per 128 B", "4 (1) + scalar prefetch, 1 per 64 B", "5 (1) with temporal vector loads",
                          "6 (1) + a fifth wave touching the next block (scalar, 128 B)", "7 (6) with one scalar load per 64 B",
                          "8 stream from a 96 MB buffer (Infinity Cache)", "9 (1) buffer loads sc1 nt", "10 (1) buffer loads sc0 sc1 nt", "11 (1) buffer loads sc1", "12 (1) with 16-byte loads, 8 consecutive entries per lane",
                          "13 (1) with one far-apart sequential stream per resident wave (sweep pattern)"};
  for (int rep = 0; rep < 2; ++rep) {
    float ms[14];
    ms[0] = time_it([&] { hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(TPB), 0, 0, col, val, x, out, nblk, window, small_blocks, remap); });
    ms[1] = time_it([&] { hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(TPB), 0, 0, col, val, x, out, nblk, window, small_blocks, remap); });
    ms[2] = time_it([&] { hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(TPB), 0, 0, col, val, x, out, nblk, window, small_blocks, remap); });
    ms[3] = time_it([&] { hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(TPB), 0, 0, col, val, x, out, nblk, window, small_blocks, remap); });
    ms[4] = time_it([&] { hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(TPB), 0, 0, col, val, x, out, nblk, window, small_blocks, remap); });
    ms[5] = time_it([&] { hipLaunchKernelGGL(probe<5>, dim3(grid), dim3(TPB), 0, 0, col, val, x, out, nblk, window, small_blocks, remap); });
    ms[6] = time_it([&] { hipLaunchKernelGGL(probe<6>, dim3(grid), dim3(TPB + WAVE), 0, 0, col, val, x, out, nblk, window, small_blocks, remap); });
    ms[7] = time_it([&] { hipLaunchKernelGGL(probe<7>, dim3(grid), dim3(TPB + WAVE), 0, 0, col, val, x, out, nblk, window, small_blocks, remap); });
    ms[8] = time_it([&] { hipLaunchKernelGGL(probe<8>, dim3(grid), dim3(TPB), 0, 0, col, val, x, out, nblk, window, small_blocks, remap); });
    ms[9] = time_it([&] { hipLaunchKernelGGL(probe<9>, dim3(grid), dim3(TPB), 0, 0, col, val, x, out, nblk, window, small_blocks, remap); });
    ms[10] = time_it([&] { hipLaunchKernelGGL(probe<10>, dim3(grid), dim3(TPB), 0, 0, col, val, x, out, nblk, window, small_blocks, remap); });
    ms[11] = time_it([&] { hipLaunchKernelGGL(probe<11>, dim3(grid), dim3(TPB), 0, 0, col, val, x, out, nblk, window, small_blocks, remap); });
    ms[12] = time_it([&] { hipLaunchKernelGGL(probe<12>, dim3(grid), dim3(TPB), 0, 0, col, val, x, out, nblk, window, small_blocks, remap); });
    ms[13] = time_it([&] { hipLaunchKernelGGL(probe<13>, dim3(grid), dim3(TPB), 0, 0, col, val, x, out, nblk, window, small_blocks, remap); });
    {
      // the same gather indices read level-major: entry k of the flat arrays IS (slice, level, lane) = (k / 640, (k % 640) / 64, k % 64)
      static double *yv = nullptr, *bv = nullptr, *yn = nullptr;
      const long nslices = count / (WAVE * ROWLEN) / 4 * 4;
      if (!yv) { CK(hipMalloc(&yv, nslices * WAVE * 8)); CK(hipMalloc(&bv, nslices * WAVE * 8)); CK(hipMalloc(&yn, nslices * WAVE * 8));
                 CK(hipMemset(yv, 0, nslices * WAVE * 8)); CK(hipMemset(bv, 0, nslices * WAVE * 8)); }
      const float t = time_it([&] { hipLaunchKernelGGL(sjds_probe, dim3(grid), dim3(TPB), 0, 0, col, val, x, yv, bv, yn, out, nslices, remap); });
      printf("  %-58s %8.4f ms  %6.1f G gathers/s\n", "14 sliced jagged miniature (rows of 10, + y, b, y')", t, nslices * WAVE * ROWLEN / (t * 1e-3) / 1e9);
    }
    for (int v = 0; v < 14; ++v) printf("  %-58s %8.4f ms  %6.1f G gathers/s\n", names[v], ms[v], count / (ms[v] * 1e-3) / 1e9);
  }
  return 0;
}
