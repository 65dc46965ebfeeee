// binned_probe.hip -- dev probe (round 4): can a TWO-PHASE product with every gather and every
// scatter served by LDS beat the L2-tiled sweep on config S (10M x 10M, 10 per row, uniform)?
//
//   phase 1 "expand":  a workgroup holds one COLUMN TILE of the gathered vector in LDS and streams the
//                      tile's entries (value 8 B + tile-local column 2 B), writing the products to P in
//                      the same order (perfectly coalesced both ways).  Entries are sorted by
//                      (super-row, tile, bin, row, column).
//   phase 2 "reduce":  a workgroup owns one BIN of consecutive rows whose products all fit in LDS; it
//                      fetches the bin's products from P (runs of one (tile, bin) cell each), parks each at
//                      its CSR position in LDS, then one lane per row adds that row's products left to
//                      right -- ascending column order, i.e. the bits of the sequential CPU loop.
//
// Also: does the 256 MiB Infinity Cache absorb a write -> read hand-off (P in super-row pieces)?
//
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o tools/binned_probe tools/binned_probe.hip
// run:   tools/binned_probe [n=10000000] [per_row=10] [CB=8192] [RB=896] [S=1] [variant=A|B] [reps=20]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static inline uint64_t splitmix(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

constexpr int WAVE = 64;
typedef double dbl4_t __attribute__((ext_vector_type(4)));
typedef unsigned short us4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ reference: plain CSR, one lane per row
__global__ void csr_ref_kernel(int m, const int *rowptr, const int *col, const double *val, const double *x, double *y) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= m) return;
  double s = 0.0;
  for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) s = s + val[k] * x[col[k]];
  y[r] = s;
}

// ------------------------------------------------------------------ phase 1
// The launch covers the padded entry range [e_begin, e_end) (multiples of 4); tile_ptr[j] = first entry of tile j
// inside this launch's range (ntiles + 1 values, multiples of 4).  Every lane handles 4 consecutive entries per step.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void expand_kernel(const double *__restrict__ val, const unsigned short *__restrict__ col,
                                                         const double *__restrict__ x, double *__restrict__ P,
                                                         const int *__restrict__ tile_ptr, int ntiles, int CB, int n) {
  extern __shared__ double xs[];
  const int tid = threadIdx.x;
  const int e_begin = tile_ptr[0], e_end = tile_ptr[ntiles];
  const int64_t total4 = ((int64_t)e_end - e_begin) / 4;
  const int64_t per4 = (total4 + gridDim.x - 1) / gridDim.x;
  const int my0 = e_begin + (int)(per4 * blockIdx.x < total4 ? per4 * blockIdx.x : total4) * 4;
  const int my1 = e_begin + (int)(per4 * (blockIdx.x + 1) < total4 ? per4 * (blockIdx.x + 1) : total4) * 4;
  if (my0 >= my1) return;
  // first tile whose range holds my0: largest j with tile_ptr[j] <= my0
  int lo = 0, hi = ntiles;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_ptr[mid] <= my0) lo = mid; else hi = mid;
  }
  int j = lo;
  int e = my0;
  while (e < my1) {
    while (tile_ptr[j + 1] <= e) ++j;            // skip empty tiles
    const int tend = min(my1, tile_ptr[j + 1]);
    __syncthreads();
    {
      const int c0 = j * CB;
      const int w = min(CB, n - c0);
      const double2 *src = reinterpret_cast<const double2 *>(x + c0);     // CB is a multiple of 2, x is padded
      double2 *dst = reinterpret_cast<double2 *>(xs);
      for (int c = tid; c < (w + 1) / 2; c += THREADS) dst[c] = src[c];
    }
    __syncthreads();
    constexpr int U = 2;                        // 2 x 4 entries per lane in flight
    for (int k = e + 4 * tid; k < tend; k += 4 * THREADS * U) {
      dbl4_t v[U];
      us4_t c[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int kk = k + u * 4 * THREADS;
        if (kk < tend) {
          v[u] = __builtin_nontemporal_load(reinterpret_cast<const dbl4_t *>(val + kk));
          c[u] = __builtin_nontemporal_load(reinterpret_cast<const us4_t *>(col + kk));
        } else {
          v[u] = dbl4_t{0, 0, 0, 0};
          c[u] = us4_t{0, 0, 0, 0};
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int kk = k + u * 4 * THREADS;
        if (kk < tend) {
          dbl4_t p;
          p.x = v[u].x * xs[c[u].x];
          p.y = v[u].y * xs[c[u].y];
          p.z = v[u].z * xs[c[u].z];
          p.w = v[u].w * xs[c[u].w];
          *reinterpret_cast<dbl4_t *>(P + kk) = p;
        }
      }
    }
    e = tend;
  }
}

// ------------------------------------------------------------------ phase 2, variant A: explicit 4-byte source index
template <int THREADS>
__global__ __launch_bounds__(THREADS) void reduce_kernel_a(const double *__restrict__ P, const unsigned *__restrict__ src,
                                                           const unsigned short *__restrict__ dst, const int *__restrict__ rowptr,
                                                           int bin0, int nbins, int per_xcd, int RB, int m, double *__restrict__ y) {
  extern __shared__ double ls[];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int bl = (b & 7) * per_xcd + (b >> 3);
  if ((b >> 3) >= per_xcd || bl >= nbins) return;
  const int bin = bin0 + bl;
  const int r0 = bin * RB, r1 = min(m, r0 + RB);
  const int k0 = rowptr[r0], k1 = rowptr[r1];
  const int ne = k1 - k0;
  constexpr int U = 8;
  for (int q = tid; q < ne; q += THREADS * U) {
    unsigned s[U];
    unsigned short d[U];
    double p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int qq = q + u * THREADS;
      const bool ok = qq < ne;
      s[u] = ok ? __builtin_nontemporal_load(src + k0 + qq) : 0u;
      d[u] = ok ? __builtin_nontemporal_load(dst + k0 + qq) : (unsigned short)0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) p[u] = (q + u * THREADS < ne) ? P[s[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) if (q + u * THREADS < ne) ls[d[u]] = p[u];
  }
  __syncthreads();
  for (int r = r0 + tid; r < r1; r += THREADS) {
    const int ks = rowptr[r] - k0, ke = rowptr[r + 1] - k0;
    double s = 0.0;
    for (int k = ks; k < ke; ++k) s = s + ls[k];
    y[r] = s;
  }
}

// ------------------------------------------------------------------ phase 2, variant B: per-chunk boundary masks
// chunk c of a bin covers stream positions [64 c, 64 c + 64); cmask bit l: entry l starts a new cell; cbase: ordinal of
// the cell of entry 0 (minus 1 when bit 0 is set); delta[ordinal]: P index of the cell's first entry minus the
// stream position of that entry.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void reduce_kernel_b(const double *__restrict__ P, const unsigned long long *__restrict__ cmask,
                                                           const int *__restrict__ cbase, const int *__restrict__ chunk_ptr,
                                                           const int *__restrict__ delta, const int *__restrict__ delta_ptr,
                                                           const unsigned short *__restrict__ dst, const int *__restrict__ rowptr,
                                                           int bin0, int nbins, int per_xcd, int RB, int m, int lds_entries,
                                                           double *__restrict__ y) {
  extern __shared__ double ls[];
  int *dl = reinterpret_cast<int *>(ls + lds_entries);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.x;
  const int bl = (b & 7) * per_xcd + (b >> 3);
  if ((b >> 3) >= per_xcd || bl >= nbins) return;
  const int bin = bin0 + bl;
  const int r0 = bin * RB, r1 = min(m, r0 + RB);
  const int k0 = rowptr[r0], k1 = rowptr[r1];
  const int ne = k1 - k0;
  const int d0 = delta_ptr[bin], d1 = delta_ptr[bin + 1];
  for (int i = tid; i < d1 - d0; i += THREADS) dl[i] = delta[d0 + i];
  __syncthreads();
  const int c0 = chunk_ptr[bin];
  const int nchunks = (ne + 63) >> 6;
  constexpr int NW = THREADS / WAVE;
  constexpr int U = 4;
  for (int c = wid; c < nchunks; c += NW * U) {
    int sidx[U];
    unsigned short d[U];
    double p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int cc = c + u * NW;
      const bool okc = cc < nchunks;
      const unsigned long long mk = okc ? cmask[c0 + cc] : 0ull;
      const int base = okc ? cbase[c0 + cc] : 0;
      const int q = cc * 64 + lane;
      const int ord = base + __popcll(mk & ((2ull << lane) - 1ull));
      const bool ok = okc && q < ne;
      sidx[u] = ok ? q + dl[ord] : 0;
      d[u] = ok ? __builtin_nontemporal_load(dst + k0 + q) : (unsigned short)0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) p[u] = P[sidx[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = (c + u * NW) * 64 + lane;
      if (c + u * NW < nchunks && q < ne) ls[d[u]] = p[u];
    }
  }
  __syncthreads();
  for (int r = r0 + tid; r < r1; r += THREADS) {
    const int ks = rowptr[r] - k0, ke = rowptr[r + 1] - k0;
    double s = 0.0;
    for (int k = ks; k < ke; ++k) s = s + ls[k];
    y[r] = s;
  }
}

// ------------------------------------------------------------------ Infinity Cache hand-off probe
__global__ void fill_kernel(double4 *buf, int64_t n4, double v) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    buf[i] = make_double4(v, v, v, v);
}
__global__ void sum_kernel(const double4 *buf, int64_t n4, double *out) {
  double s = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const double4 t = buf[i];
    s += t.x + t.y + t.z + t.w;
  }
  if (s == 12345.678) out[0] = s;
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 10000000;
  const int per_row = argc > 2 ? atoi(argv[2]) : 10;
  const int CB = argc > 3 ? atoi(argv[3]) : 8192;
  const int RB = argc > 4 ? atoi(argv[4]) : 896;
  const int S = argc > 5 ? atoi(argv[5]) : 1;
  const char variant = argc > 6 ? argv[6][0] : 'A';
  const int reps = argc > 7 ? atoi(argv[7]) : 20;
  const int m = n;
  const int64_t nnz = (int64_t)m * per_row;
  printf("binned_probe: m = n = %d, %d per row (nnz %lld), CB %d cols (%d KB of LDS), RB %d rows, S %d super-rows, variant %c\n",
         n, per_row, (long long)nnz, CB, CB * 8 / 1024, RB, S, variant);

  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  // ---- Infinity Cache hand-off: write N bytes, read them back, alternate
  if (getenv("PROBE_MALL")) {
    for (int mb : {32, 64, 96, 128, 192, 256, 512, 1024, 2048}) {
      const int64_t n4 = (int64_t)mb * 1048576 / 32;
      double4 *buf;
      double *out;
      CK(hipMalloc((void **)&buf, n4 * 32));
      CK(hipMalloc((void **)&out, 8));
      for (int w = 0; w < 2; ++w) {
        fill_kernel<<<2048, 256, 0, st>>>(buf, n4, 1.0);
        sum_kernel<<<2048, 256, 0, st>>>(buf, n4, out);
      }
      CK(hipStreamSynchronize(st));
      const int R = 10;
      float ms_w = 0, ms_r = 0;
      for (int r = 0; r < R; ++r) {
        float t;
        CK(hipEventRecord(e0, st));
        fill_kernel<<<2048, 256, 0, st>>>(buf, n4, 1.0 + r);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t, e0, e1));
        ms_w += t;
        CK(hipEventRecord(e0, st));
        sum_kernel<<<2048, 256, 0, st>>>(buf, n4, out);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t, e0, e1));
        ms_r += t;
      }
      printf("  hand-off %5d MB: write %.1f GB/s, read-back %.1f GB/s\n", mb, mb / 1024.0 * 1.073741824 / (ms_w / R) * 1e3,
             mb / 1024.0 * 1.073741824 / (ms_r / R) * 1e3);
      CK(hipFree(buf));
      CK(hipFree(out));
    }
  }

  // ---- the matrix (rows sorted by column), CSR
  auto t_host = std::chrono::steady_clock::now();
  std::vector<int> rowptr((size_t)m + 1);
  std::vector<int> col((size_t)nnz);
  std::vector<double> val((size_t)nnz);
  {
    uint64_t seed = 12345;
    for (int r = 0; r < m; ++r) {
      rowptr[r] = r * per_row;
      int *c = col.data() + (size_t)r * per_row;
      for (int k = 0; k < per_row; ++k) {
        for (;;) {
          const int cand = (int)(splitmix(seed) % (uint64_t)n);
          bool dup = false;
          for (int q = 0; q < k; ++q) dup |= c[q] == cand;
          if (!dup) { c[k] = cand; break; }
        }
      }
      std::sort(c, c + per_row);
      for (int k = 0; k < per_row; ++k) val[(size_t)r * per_row + k] = (double)(int64_t)(splitmix(seed) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
    rowptr[m] = (int)nnz;
  }
  std::vector<double> x((size_t)n + 16, 0.0);
  {
    uint64_t seed = 777;
    for (int i = 0; i < n; ++i) x[i] = (double)(int64_t)(splitmix(seed) >> 11) * (1.0 / 9007199254740992.0);
  }
  const int ntiles = (n + CB - 1) / CB;
  const int nbins = (m + RB - 1) / RB;
  const int bins_per_s = (nbins + S - 1) / S;
  // cells in (super-row, tile, bin) order
  const size_t ncells = (size_t)S * ntiles * bins_per_s;
  std::vector<int> cell_cnt(ncells + 1, 0);
  auto cell_of = [&](int r, int c) -> size_t {
    const int bin = r / RB, s = bin / bins_per_s, j = c / CB;
    return ((size_t)s * ntiles + j) * bins_per_s + (bin - s * bins_per_s);
  };
  for (int r = 0; r < m; ++r)
    for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) cell_cnt[cell_of(r, col[k])] += 1;
  // P offsets: every (super-row, tile) segment padded to a multiple of 4 entries
  std::vector<int> cell_start(ncells + 1);
  std::vector<int> seg_ptr((size_t)S * (ntiles + 1));
  int64_t ptotal = 0;
  for (int s = 0; s < S; ++s) {
    for (int j = 0; j < ntiles; ++j) {
      seg_ptr[(size_t)s * (ntiles + 1) + j] = (int)ptotal;
      for (int bl = 0; bl < bins_per_s; ++bl) {
        const size_t cidx = ((size_t)s * ntiles + j) * bins_per_s + bl;
        cell_start[cidx] = (int)ptotal;
        ptotal += cell_cnt[cidx];
      }
      ptotal = (ptotal + 3) & ~3LL;
    }
    seg_ptr[(size_t)s * (ntiles + 1) + ntiles] = (int)ptotal;
  }
  std::vector<double> val1((size_t)ptotal, 0.0);
  std::vector<unsigned short> col1((size_t)ptotal, 0);
  std::vector<unsigned> src((size_t)nnz);
  std::vector<unsigned short> dst((size_t)nnz);
  {
    // stream position of a cell inside its bin: prefix over tiles
    std::vector<int> bin_off(ncells);
    for (int s = 0; s < S; ++s)
      for (int bl = 0; bl < bins_per_s; ++bl) {
        int run = 0;
        for (int j = 0; j < ntiles; ++j) {
          const size_t cidx = ((size_t)s * ntiles + j) * bins_per_s + bl;
          bin_off[cidx] = run;
          run += cell_cnt[cidx];
        }
      }
    std::vector<int> fill(ncells, 0);
    for (int r = 0; r < m; ++r) {
      const int bin = r / RB;
      const int kb = rowptr[bin * RB];
      for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const size_t cidx = cell_of(r, col[k]);
        const int t = fill[cidx]++;
        const int pe = cell_start[cidx] + t;
        val1[(size_t)pe] = val[(size_t)k];
        col1[(size_t)pe] = (unsigned short)(col[k] % CB);
        const int q = kb + bin_off[cidx] + t;
        src[(size_t)q] = (unsigned)pe;
        dst[(size_t)q] = (unsigned short)(k - kb);
      }
    }
  }
  // variant B tables
  std::vector<unsigned long long> cmask;
  std::vector<int> cbase, chunk_ptr((size_t)nbins + 1, 0), delta, delta_ptr((size_t)nbins + 1, 0);
  int max_cells_per_bin = 0;
  for (int bin = 0; bin < nbins; ++bin) {
    const int s = bin / bins_per_s, bl = bin - s * bins_per_s;
    const int r0 = bin * RB, r1 = std::min(m, r0 + RB);
    const int ne = rowptr[r1] - rowptr[r0];
    const int nch = (ne + 63) / 64;
    chunk_ptr[bin + 1] = chunk_ptr[bin] + nch;
    const size_t cb0 = cmask.size();
    cmask.resize(cb0 + nch, 0ull);
    cbase.resize(cb0 + nch, 0);
    int run = 0, ord = 0;
    for (int j = 0; j < ntiles; ++j) {
      const size_t cidx = ((size_t)s * ntiles + j) * bins_per_s + bl;
      const int cnt = cell_cnt[cidx];
      if (!cnt) continue;
      delta.push_back(cell_start[cidx] - run);
      cmask[cb0 + run / 64] |= 1ull << (run & 63);
      // chunks whose entry 0 lies in this cell
      for (int c = (run + 63) / 64; c * 64 < run + cnt; ++c) cbase[cb0 + c] = (c * 64 == run) ? ord - 1 : ord;
      run += cnt;
      ++ord;
    }
    delta_ptr[bin + 1] = (int)delta.size();
    max_cells_per_bin = std::max(max_cells_per_bin, ord);
  }
  printf("host build %.1f s; %d tiles x %d bins, %.1f entries per cell, P holds %lld entries; meta B: %.2f B/nnz\n",
         std::chrono::duration<double>(std::chrono::steady_clock::now() - t_host).count(), ntiles, nbins,
         (double)nnz / ((double)ntiles * nbins), (long long)ptotal,
         (double)(cmask.size() * 12 + delta.size() * 4) / (double)nnz);

  // ---- device
  int *d_rowptr, *d_col, *d_seg, *d_cbase, *d_chunk_ptr, *d_delta, *d_delta_ptr;
  double *d_val, *d_x, *d_y, *d_yref, *d_val1, *d_P;
  unsigned short *d_col1, *d_dst;
  unsigned *d_src;
  unsigned long long *d_cmask;
#define UP(dptr, vec) do { CK(hipMalloc((void **)&dptr, std::max<size_t>(1, (vec).size()) * sizeof((vec)[0]))); \
    CK(hipMemcpy(dptr, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice)); } while (0)
  UP(d_rowptr, rowptr); UP(d_col, col); UP(d_val, val); UP(d_x, x); UP(d_seg, seg_ptr);
  UP(d_val1, val1); UP(d_col1, col1); UP(d_src, src); UP(d_dst, dst);
  UP(d_cmask, cmask); UP(d_cbase, cbase); UP(d_chunk_ptr, chunk_ptr); UP(d_delta, delta); UP(d_delta_ptr, delta_ptr);
  CK(hipMalloc((void **)&d_y, sizeof(double) * (size_t)m));
  CK(hipMalloc((void **)&d_yref, sizeof(double) * (size_t)m));
  CK(hipMalloc((void **)&d_P, sizeof(double) * (size_t)ptotal));
  CK(hipMemset(d_y, 0, sizeof(double) * (size_t)m));

  csr_ref_kernel<<<(m + 255) / 256, 256, 0, st>>>(m, d_rowptr, d_col, d_val, d_x, d_yref);
  CK(hipStreamSynchronize(st));

  constexpr int T1 = 512, T2 = 256;
  const size_t lds1 = (size_t)CB * 8;
  int max_ne = 0;
  for (int bin = 0; bin < nbins; ++bin) max_ne = std::max(max_ne, rowptr[std::min(m, (bin + 1) * RB)] - rowptr[bin * RB]);
  const size_t lds2a = (size_t)max_ne * 8;
  const size_t lds2b = (size_t)max_ne * 8 + (size_t)max_cells_per_bin * 4 + 16;
  CK(hipFuncSetAttribute((const void *)expand_kernel<T1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
  CK(hipFuncSetAttribute((const void *)reduce_kernel_a<T2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2a));
  CK(hipFuncSetAttribute((const void *)reduce_kernel_b<T2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2b));
  const int g1 = getenv("PROBE_G1") ? atoi(getenv("PROBE_G1")) : 256 * (int)std::max<size_t>(1, (160 * 1024) / (lds1 + 1024));
  printf("phase 1: %d workgroups of %d threads per launch, %zu B of LDS; phase 2: %zu B of LDS (max %d entries per bin)\n",
         g1, T1, lds1, variant == 'A' ? lds2a : lds2b, max_ne);

  auto phase1 = [&](int s) {
    expand_kernel<T1><<<g1, T1, lds1, st>>>(d_val1, d_col1, d_x, d_P, d_seg + (size_t)s * (ntiles + 1), ntiles, CB, n);
  };
  auto phase2 = [&](int s) {
    const int b0 = s * bins_per_s, nb = std::min(bins_per_s, nbins - b0);
    if (nb <= 0) return;
    const int per_xcd = (nb + 7) / 8;
    if (variant == 'A')
      reduce_kernel_a<T2><<<per_xcd * 8, T2, lds2a, st>>>(d_P, d_src, d_dst, d_rowptr, b0, nb, per_xcd, RB, m, d_y);
    else
      reduce_kernel_b<T2><<<per_xcd * 8, T2, lds2b, st>>>(d_P, d_cmask, d_cbase, d_chunk_ptr, d_delta, d_delta_ptr, d_dst, d_rowptr,
                                                         b0, nb, per_xcd, RB, m, max_ne, d_y);
  };
  auto product = [&]() { for (int s = 0; s < S; ++s) { phase1(s); phase2(s); } };

  product();
  CK(hipStreamSynchronize(st));
  CK(hipGetLastError());
  {
    std::vector<double> y((size_t)m), yr((size_t)m);
    CK(hipMemcpy(y.data(), d_y, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost));
    CK(hipMemcpy(yr.data(), d_yref, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost));
    int64_t bad = 0;
    for (int r = 0; r < m; ++r) bad += memcmp(&y[r], &yr[r], 8) != 0;
    printf("bitwise check against the one-lane-per-row CSR kernel: %lld rows differ of %d\n", (long long)bad, m);
  }
  auto time_it = [&](const char *name, auto f, double bytes) {
    for (int w = 0; w < 3; ++w) f();
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("  %-28s %.4f ms   %.0f GB/s of %s\n", name, ms, bytes / ms * 1e-6, "its own traffic");
    return ms;
  };
  const double b1 = (double)ptotal * 18.0 + 8.0 * n;
  const double b2 = (double)nnz * (variant == 'A' ? 14.0 : 10.0) + (variant == 'A' ? 0.0 : (double)(cmask.size() * 12 + delta.size() * 4)) + 12.0 * m;
  time_it("phase 1 (all super-rows)", [&] { for (int s = 0; s < S; ++s) phase1(s); }, b1);
  time_it("phase 2 (all super-rows)", [&] { for (int s = 0; s < S; ++s) phase2(s); }, b2);
  const float ms = time_it("product (interleaved)", product, b1 + b2);
  const double alg = 12.0 * nnz + 4.0 * (m + 1) + 8.0 * n + 8.0 * m;
  printf("product: %.4f ms; algorithmic bytes %.3f GB -> %.0f GB/s = %.3f of 8 TB/s (bare product, no fused epilogue)\n", ms, alg * 1e-9,
         alg / ms * 1e-6, alg / ms * 1e-6 / 8000.0);
  return 0;
}
