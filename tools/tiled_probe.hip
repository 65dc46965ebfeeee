// Dev probe: feasibility of a cache-blocked SpMV.  A workgroup owns a row
// block (accumulators in LDS) and sweeps column tiles in order, so that all
// resident workgroups gather from the same x tile (L2-resident) at about the
// same time.  Synthetic uniform cells: every (row block, tile) cell has the
// same nnz.  Usage: tiled_probe n_millions nnz_millions tile_cols rows_per_block threads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int THREADS, int U>
__global__ __launch_bounds__(THREADS) void tiled_kernel(const unsigned *__restrict__ idx, const double *__restrict__ val,
                                                        const double *__restrict__ x, double *__restrict__ out,
                                                        int ntiles, int tile_shift, int rows_per_block, long cell_nnz) {
  extern __shared__ double acc[];
  for (int r = threadIdx.x; r < rows_per_block; r += THREADS) acc[r] = 0.0;
  __syncthreads();
  const long base = (long)blockIdx.x * ntiles * cell_nnz;
  const unsigned cmask = (1u << tile_shift) - 1u;
  for (int t = 0; t < ntiles; ++t) {
    const long cb = base + (long)t * cell_nnz;
    const double *xt = x + ((long)t << tile_shift);
    for (long k0 = threadIdx.x; k0 < cell_nnz; k0 += (long)THREADS * U) {
      unsigned p[U]; double v[U]; double xv[U];
#pragma unroll
      for (int i = 0; i < U; ++i) { long k = k0 + (long)i * THREADS; bool ok = k < cell_nnz; p[i] = ok ? __builtin_nontemporal_load(idx + cb + k) : 0u; v[i] = ok ? __builtin_nontemporal_load(val + cb + k) : 0.0; }
#pragma unroll
      for (int i = 0; i < U; ++i) xv[i] = xt[p[i] & cmask];
#pragma unroll
      for (int i = 0; i < U; ++i) { long k = k0 + (long)i * THREADS; if (k < cell_nnz) { unsigned r = p[i] >> tile_shift; acc[r] += v[i] * xv[i]; } }
    }
    __syncthreads();
  }
  for (int r = threadIdx.x; r < rows_per_block; r += THREADS) out[(long)blockIdx.x * rows_per_block + r] = acc[r];
}

int main(int argc, char **argv) {
  long n = (argc > 1 ? atol(argv[1]) : 10) * 1000000L;
  long nnz = (argc > 2 ? atol(argv[2]) : 100) * 1000000L;
  int tile_shift = argc > 3 ? atoi(argv[3]) : 18;
  int rpb = argc > 4 ? atoi(argv[4]) : 8192;
  int threads = argc > 5 ? atoi(argv[5]) : 512;
  long tile = 1L << tile_shift;
  int ntiles = (int)((n + tile - 1) / tile);
  long m = n;
  int nrb = (int)((m + rpb - 1) / rpb);
  long cell_nnz = nnz / ((long)nrb * ntiles);
  long total = cell_nnz * nrb * ntiles;
  if ((32 - tile_shift) < 1 || (rpb > (1 << (32 - tile_shift)))) { printf("rows_per_block %d does not fit in %d bits\n", rpb, 32 - tile_shift); return 1; }
  std::vector<unsigned> h(total);
  std::mt19937_64 rng(1);
  for (long i = 0; i < total; ++i) { unsigned c = (unsigned)(rng() % tile), r = (unsigned)(rng() % rpb); h[i] = (r << tile_shift) | c; }
  unsigned *idx; double *val, *x, *out;
  CK(hipMalloc(&idx, total * 4)); CK(hipMemcpy(idx, h.data(), total * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&val, total * 8)); CK(hipMemset(val, 0, total * 8));
  CK(hipMalloc(&x, (long)ntiles * tile * 8)); CK(hipMemset(x, 0, (long)ntiles * tile * 8));
  CK(hipMalloc(&out, (long)nrb * rpb * 8));
  size_t lds = (size_t)rpb * 8;
  auto run = [&](int thr) {
#define LAUNCH(T, UU) do { CK(hipFuncSetAttribute((const void *)tiled_kernel<T, UU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      hipLaunchKernelGGL((tiled_kernel<T, UU>), dim3(nrb), dim3(T), lds, 0, idx, val, x, out, ntiles, tile_shift, rpb, cell_nnz); } while (0)
    if (thr == 256) LAUNCH(256, 4); else if (thr == 512) LAUNCH(512, 4); else LAUNCH(1024, 4);
  };
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  run(threads); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); run(threads); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
  printf("n=%ld nnz=%ld tile=%ld cols (%d tiles) rows/block=%d (%d blocks, LDS %zu KB) threads=%d cell_nnz=%ld : %.3f ms  %.1f Gnnz/s  (12B/nnz -> %.0f GB/s)\n",
         n, total, tile, ntiles, rpb, nrb, lds / 1024, threads, cell_nnz, best, total / best / 1e6, total * 12.0 / best / 1e6);
  return 0;
}
