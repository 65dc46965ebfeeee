// device_layout.hpp -- part of the single translation unit pdhg_hip.hip (included there, after layout.hpp).
// Layout construction ON THE DEVICE for large matrices: the caller's CSC arrays are uploaded
// as they are and everything else happens in HBM --
//   narrowing to 32-bit 0-based CSR(A')                      (elementwise, validated)
//   CSR(A) by a stable LSD radix sort of the entries by row  (8-bit digits; entries arrive in
//       column-major order, so a STABLE sort by row leaves every row in ascending column order:
//       exactly what the host's bucket sort produces)
//   the tiled sweep's tile-major copy by a stable counting sort per wave (peer masks from
//       bit ballots; ranks by lane order)
// -- and only the small tables (row pointers, per-(wave, tile) counts) travel back for the
// serial / greedy parts of the plan (row blocks, wave row ranges, step lists), which stay on the
// host in the same code the host builders use.  The result is bit-identical to the host
// builders' (tests/test_gpu_device_layout.py compares checksums of every device array).
// The host builders took 1.5 s inside the library at config S, bound by host memory bandwidth
// (profiles/r03_create.txt); a rescale-heavy or re-solve workflow pays that at every create.
#pragma once

namespace {

// ---------------------------------------------------------------- scan

constexpr int SCAN_TPB = 256, SCAN_ITEMS = 8, SCAN_TILE = SCAN_TPB * SCAN_ITEMS;

// out[i] = sum of in[0..i) within each tile of SCAN_TILE elements; tile_sum[b] = the tile's total
__global__ __launch_bounds__(SCAN_TPB) void scan_tiles_kernel(const int *__restrict__ in, int *__restrict__ out,
                                                              int *__restrict__ tile_sum, int64_t n) {
  __shared__ int wsum[SCAN_TPB / WAVE];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS], run = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    const int64_t k = base + i;
    const int t = k < n ? in[k] : 0;
    v[i] = run;
    run += t;
  }
  // exclusive scan of the per-thread totals: wave shuffles, then the wave totals
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  int incl = run;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const int t = __shfl_up(incl, d, WAVE);
    if (lane >= d) incl += t;
  }
  if (lane == WAVE - 1) wsum[wid] = incl;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wid; ++w) woff += wsum[w];
  const int excl = woff + incl - run;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    const int64_t k = base + i;
    if (k < n) out[k] = excl + v[i];
  }
  if (threadIdx.x == SCAN_TPB - 1 && tile_sum) tile_sum[blockIdx.x] = woff + incl;
}

__global__ __launch_bounds__(SCAN_TPB) void scan_add_kernel(int *__restrict__ out, const int *__restrict__ tile_off, int64_t n) {
  const int add = tile_off[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    const int64_t k = base + (int64_t)i * SCAN_TPB;
    if (k < n) out[k] += add;
  }
}

// exclusive prefix sum of n ints (in may equal out); `total` (device, optional) receives the sum
int device_exclusive_scan(const int *in, int *out, int64_t n, int *total, hipStream_t st) {
  if (n <= 0) { if (total) HIP_TRY(hipMemsetAsync(total, 0, sizeof(int), st)); return 0; }
  const int64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  int *sums = nullptr;
  HIP_TRY(hipMalloc((void **)&sums, sizeof(int) * (size_t)(tiles + 1)));
  hipLaunchKernelGGL(scan_tiles_kernel, dim3((unsigned)tiles), dim3(SCAN_TPB), 0, st, in, out, sums, n);
  int rc = 0;
  if (tiles > 1) {
    rc = device_exclusive_scan(sums, sums, tiles, total, st);
    if (!rc) hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)tiles), dim3(SCAN_TPB), 0, st, out, (const int *)sums, n);
  } else if (total) {
    HIP_TRY(hipMemcpyAsync(total, sums, sizeof(int), hipMemcpyDeviceToDevice, st));
  }
  hipError_t e = hipGetLastError();
  HIP_TRY(hipStreamSynchronize(st));
  (void)hipFree(sums);
  if (rc) return rc;
  HIP_TRY(e);
  return 0;
}

// ---------------------------------------------------------------- column slabs of a stream layout

// cnt[r] = entries of row r with c0 <= column < c1; rows that are long in the full matrix (the long-row
// path's) and the extra slot cnt[rows] count 0, so that the exclusive scan of cnt[0..rows] is the slab's
// row-pointer array.  One thread per row: a one-time pass, rows are short here by construction.
__global__ __launch_bounds__(TPB) void slab_count_kernel(int rows, const int *__restrict__ rowptr,
                                                         const int *__restrict__ col, int c0, int c1, int max_row,
                                                         int *__restrict__ cnt) {
  for (int r = blockIdx.x * TPB + threadIdx.x; r <= rows; r += gridDim.x * TPB) {
    int c = 0;
    if (r < rows) {
      const int b = rowptr[r], e = rowptr[r + 1];
      if (e - b <= max_row)
        for (int k = b; k < e; ++k) { const int j = col[k]; c += (j >= c0 && j < c1); }
    }
    cnt[r] = c;
  }
}

// the slab's entries, every row in its original order (what the host builder's loop writes)
__global__ __launch_bounds__(TPB) void slab_fill_kernel(int rows, const int *__restrict__ rowptr,
                                                        const int *__restrict__ col, const double *__restrict__ val,
                                                        int c0, int c1, int max_row, const int *__restrict__ slab_rowptr,
                                                        int *__restrict__ sc, double *__restrict__ sv) {
  for (int r = blockIdx.x * TPB + threadIdx.x; r < rows; r += gridDim.x * TPB) {
    const int b = rowptr[r], e = rowptr[r + 1];
    if (e - b > max_row) continue;
    int q = slab_rowptr[r];
    for (int k = b; k < e; ++k) {
      const int j = col[k];
      if (j >= c0 && j < c1) { sc[q] = j; sv[q] = val[k]; ++q; }
    }
  }
}

// ---------------------------------------------------------------- peer masks

// lanes of the wave whose `key` equals this lane's, among the lanes in `valid`: one ballot per key bit
__device__ __forceinline__ unsigned long long peer_mask(unsigned key, int bits, unsigned long long valid) {
  unsigned long long m = valid;
  for (int b = 0; b < bits; ++b) {
    const unsigned long long vote = __ballot((key >> b) & 1u);
    m &= ((key >> b) & 1u) ? vote : ~vote;
  }
  return m;
}
__device__ __forceinline__ unsigned long long lanes_below(int lane) { return lane == 0 ? 0ull : (~0ull >> (WAVE - lane)); }

// ---------------------------------------------------------------- stable radix sort of (key; int, double)

constexpr int RS_TPB = 256, RS_WAVES = RS_TPB / WAVE, RS_CHUNKS = 16;      // a workgroup sorts 4 x 16 x 64 = 4096 entries
constexpr int RS_TILE = RS_TPB * RS_CHUNKS, RS_BITS = 8, RS_BINS = 1 << RS_BITS;

// digit counts of every workgroup's tile, bin-major: hist[d * nwg + wg]
__global__ __launch_bounds__(RS_TPB) void rs_hist_kernel(const int *__restrict__ key, int64_t n, int shift,
                                                         int *__restrict__ hist, int nwg) {
  __shared__ int cnt[RS_BINS];
  for (int d = threadIdx.x; d < RS_BINS; d += RS_TPB) cnt[d] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
  for (int i = 0; i < RS_CHUNKS; ++i) {
    const int64_t k = base + (int64_t)i * RS_TPB + threadIdx.x;
    if (k < n) atomicAdd(&cnt[((unsigned)key[k] >> shift) & (RS_BINS - 1)], 1);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < RS_BINS; d += RS_TPB) hist[(size_t)d * nwg + blockIdx.x] = cnt[d];
}

// Stable scatter.  Wave w of the workgroup owns the tile's entries [w * 1024, (w + 1) * 1024) and
// walks them 64 at a time in order; within a chunk the rank of an entry among its peers (same
// digit) is its position by lane, so equal keys keep their input order.
__global__ __launch_bounds__(RS_TPB) void rs_scatter_kernel(const int *__restrict__ key, const int *__restrict__ pa,
                                                            const double *__restrict__ pb, int64_t n, int shift,
                                                            const int *__restrict__ offs, int nwg,
                                                            int *__restrict__ key_out, int *__restrict__ pa_out,
                                                            double *__restrict__ pb_out) {
  __shared__ int wcnt[RS_WAVES][RS_BINS];
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  for (int d = threadIdx.x; d < RS_WAVES * RS_BINS; d += RS_TPB) (&wcnt[0][0])[d] = 0;
  __syncthreads();
  const int64_t wbase = (int64_t)blockIdx.x * RS_TILE + (int64_t)wid * (RS_CHUNKS * WAVE);
  // per-wave digit counts of the wave's own entries
  for (int i = 0; i < RS_CHUNKS; ++i) {
    const int64_t k = wbase + (int64_t)i * WAVE + lane;
    if (k < n) atomicAdd(&wcnt[wid][((unsigned)key[k] >> shift) & (RS_BINS - 1)], 1);
  }
  __syncthreads();
  // wcnt[w][d] <- global start of wave w's entries with digit d
  for (int d = threadIdx.x; d < RS_BINS; d += RS_TPB) {
    int run = offs[(size_t)d * nwg + blockIdx.x];
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) {
      const int c = wcnt[w][d];
      wcnt[w][d] = run;
      run += c;
    }
  }
  __syncthreads();
  volatile int *cur = wcnt[wid];
  for (int i = 0; i < RS_CHUNKS; ++i) {
    const int64_t k = wbase + (int64_t)i * WAVE + lane;
    const bool ok = k < n;
    const unsigned long long valid = __ballot(ok);
    if (!valid) break;
    const int kk = ok ? key[k] : 0;
    const unsigned d = ((unsigned)kk >> shift) & (RS_BINS - 1);
    const unsigned long long peers = peer_mask(d, RS_BITS, valid);
    const int rank = __popcll(peers & lanes_below(lane));
    int pos = 0;
    if (ok) pos = cur[d] + rank;
    __builtin_amdgcn_wave_barrier();
    if (ok && rank == 0) cur[d] = cur[d] + __popcll(peers);       // one lane per distinct digit
    __builtin_amdgcn_wave_barrier();
    if (ok) {
      key_out[pos] = kk;
      pa_out[pos] = pa[k];
      pb_out[pos] = pb[k];
    }
  }
}

// Sorts (key, a, b) stably by key < 2^key_bits.  The result ends in (key, a, b) or in the
// scratch triple: *in_scratch says which.
int device_radix_sort(int *key, int *a, double *b, int *key2, int *a2, double *b2, int64_t n, int key_bits,
                      bool *in_scratch, hipStream_t st) {
  *in_scratch = false;
  if (n <= 0) return 0;
  const int nwg = (int)((n + RS_TILE - 1) / RS_TILE);
  int *hist = nullptr;
  HIP_TRY(hipMalloc((void **)&hist, sizeof(int) * (size_t)RS_BINS * nwg));
  int rc = 0;
  for (int shift = 0; shift < key_bits && !rc; shift += RS_BITS) {
    hipLaunchKernelGGL(rs_hist_kernel, dim3(nwg), dim3(RS_TPB), 0, st, (const int *)key, n, shift, hist, nwg);
    rc = device_exclusive_scan(hist, hist, (int64_t)RS_BINS * nwg, nullptr, st);
    if (rc) break;
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(nwg), dim3(RS_TPB), 0, st, (const int *)key, (const int *)a, (const double *)b, n,
                       shift, (const int *)hist, nwg, key2, a2, b2);
    std::swap(key, key2); std::swap(a, a2); std::swap(b, b2);
    *in_scratch = !*in_scratch;
  }
  hipError_t e = hipGetLastError();
  hipError_t e2 = hipStreamSynchronize(st);
  (void)hipFree(hist);
  if (rc) return rc;
  HIP_TRY(e);
  HIP_TRY(e2);
  return 0;
}

// ---------------------------------------------------------------- ingest

// CSR(A') = the CSC arrays narrowed to 32-bit 0-based indices; flags[0] |= 1 on a row index out of range,
// |= 2 on a non-monotone column pointer.  Also the row of every entry as sort key and its column as payload.
__global__ __launch_bounds__(TPB) void ingest_colptr_kernel(const int64_t *__restrict__ colptr, int64_t cols, int64_t nnz,
                                                            int base, int *__restrict__ t_rowptr, int *__restrict__ flags) {
  const int64_t j = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (j > cols) return;
  const int64_t v = colptr[j] - base;
  const int64_t prev = j > 0 ? colptr[j - 1] - base : 0;
  if (v < 0 || v > nnz || v < prev) atomicOr(flags, 2);
  t_rowptr[j] = (int)v;
}

// one wave per 64 columns' worth is wasteful for short columns; one lane per ENTRY with a binary search for its
// column keeps every access coalesced
__global__ __launch_bounds__(TPB) void ingest_entries_kernel(const int64_t *__restrict__ rowval, const int *__restrict__ t_rowptr,
                                                             int64_t nnz, int64_t rows, int64_t cols, int base,
                                                             int *__restrict__ t_col, int *__restrict__ key_row,
                                                             int *__restrict__ pay_col, int *__restrict__ row_cnt,
                                                             int *__restrict__ flags) {
  const int64_t stride = (int64_t)gridDim.x * TPB;
  for (int64_t k = (int64_t)blockIdx.x * TPB + threadIdx.x; k < nnz; k += stride) {
    const int64_t r = rowval[k] - base;
    if (r < 0 || r >= rows) { atomicOr(flags, 1); continue; }
    // column of entry k: last j with t_rowptr[j] <= k
    int64_t lo = 0, hi = cols;           // invariant: t_rowptr[lo] <= k < t_rowptr[hi]
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)t_rowptr[mid] <= k) lo = mid; else hi = mid;
    }
    t_col[k] = (int)r;
    key_row[k] = (int)r;
    pay_col[k] = (int)lo;
    atomicAdd(&row_cnt[r], 1);
  }
}

// ---------------------------------------------------------------- the tiled sweep's tables

// Tile of column c: uniform width tile_cols, or (map16 != nullptr: tiles of equal nonzeros and different
// widths, build_tiled) the tile of its 16-column group.
__device__ __forceinline__ int tw_tile_of(int c, int tile_cols, const int *__restrict__ map16) {
  return map16 ? map16[c >> 4] : c / tile_cols;
}

// Per wave: entries per tile and the longest same-row run inside one tile.
// One lane per row (order is irrelevant for counts): cnt[w * ntiles + t].
__global__ __launch_bounds__(TPB) void tw_count_kernel(const int2 *__restrict__ wave_rows, int nwaves,
                                                       const int *__restrict__ rowptr, const int *__restrict__ col,
                                                       int tile_cols, const int *__restrict__ map16, int ntiles,
                                                       int *__restrict__ cnt, int *__restrict__ max_run_of_wg) {
  extern __shared__ int tw_hist[];            // [TPB / WAVE][ntiles]
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  const int w = blockIdx.x * (TPB / WAVE) + wid;
  int *h = tw_hist + (size_t)wid * ntiles;
  for (int t = lane; t < ntiles; t += WAVE) h[t] = 0;
  __builtin_amdgcn_wave_barrier();
  int max_run = 0;
  if (w < nwaves) {
    const int2 rr = wave_rows[w];
    for (int r = rr.x + lane; r < rr.y; r += WAVE) {
      int run = 0, run_tile = -1;
      for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
        const int t = tw_tile_of(col[k], tile_cols, map16);
        run = (t == run_tile) ? run + 1 : 1;
        run_tile = t;
        max_run = max(max_run, run);
        atomicAdd(&h[t], 1);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (w < nwaves)
    for (int t = lane; t < ntiles; t += WAVE) cnt[(size_t)w * ntiles + t] = h[t];
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) max_run = max(max_run, __shfl_down(max_run, off, WAVE));
  if (lane == 0 && w < nwaves) atomicMax(&max_run_of_wg[w / TW_WPB], max_run);
}

// entries per group of 16 columns over the rows that are not long (the skew test of build_tiled)
__global__ __launch_bounds__(TPB) void cnt16_kernel(int rows, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                    int *__restrict__ cnt16, int long_thr) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * TPB + threadIdx.x) / WAVE, nw = (int64_t)gridDim.x * TPB / WAVE;
  for (int64_t r = wave; r < rows; r += nw) {
    const int k0 = rowptr[r], k1 = rowptr[r + 1];
    if (k1 - k0 > long_thr) continue;
    for (int k = k0 + lane; k < k1; k += WAVE) atomicAdd(&cnt16[col[k] >> 4], 1);
  }
}

// The tile-major copy: wave w's entries [rowptr[r0], rowptr[r1]) stably counting-sorted by tile into
// pk / tv at base[w] + cell_start[w][tile] + rank.  cell_start comes in as cnt's exclusive prefix
// over tiles (computed here, in LDS).  64 entries at a time, in order; ranks by lane order.
__global__ __launch_bounds__(TPB) void tw_fill_kernel(const int2 *__restrict__ wave_rows, int nwaves,
                                                      const int64_t *__restrict__ wave_base,
                                                      const int *__restrict__ rowptr, const int *__restrict__ col,
                                                      const double *__restrict__ val, int tile_cols,
                                                      const int *__restrict__ map16, const int *__restrict__ tstart, int ntiles,
                                                      int tile_bits, int tile_shift, const int *__restrict__ cnt,
                                                      unsigned *__restrict__ pk, double *__restrict__ tv) {
  extern __shared__ int tw_cur[];             // [TPB / WAVE][ntiles]
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  const int w = blockIdx.x * (TPB / WAVE) + wid;
  if (w >= nwaves) return;
  volatile int *cur = tw_cur + (size_t)wid * ntiles;
  // exclusive prefix of this wave's counts over the tiles (wave-sequential in blocks of 64)
  int carry = 0;
  for (int t0 = 0; t0 < ntiles; t0 += WAVE) {
    const int t = t0 + lane;
    const int c = t < ntiles ? cnt[(size_t)w * ntiles + t] : 0;
    int incl = c;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const int u = __shfl_up(incl, d, WAVE);
      if (lane >= d) incl += u;
    }
    if (t < ntiles) cur[t] = carry + incl - c;
    carry += __shfl(incl, WAVE - 1, WAVE);
  }
  __builtin_amdgcn_wave_barrier();
  const int2 rr = wave_rows[w];
  const int k0 = rowptr[rr.x], k1 = rowptr[rr.y];
  const int64_t base = wave_base[w];
  for (int kb = k0; kb < k1; kb += WAVE) {
    const int k = kb + lane;
    const bool ok = k < k1;
    const unsigned long long valid = __ballot(ok);
    const int c = ok ? col[k] : 0;
    const int t = ok ? tw_tile_of(c, tile_cols, map16) : 0;
    // the entry's row: last r in [r0, r1) with rowptr[r] <= k
    int lo = rr.x, hi = rr.y;
    while (ok && hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (rowptr[mid] <= k) lo = mid; else hi = mid;
    }
    const unsigned long long peers = peer_mask((unsigned)t, tile_bits, valid);
    const int rank = __popcll(peers & lanes_below(lane));
    int pos = 0;
    if (ok) pos = cur[t] + rank;
    __builtin_amdgcn_wave_barrier();
    if (ok && rank == 0) cur[t] = cur[t] + __popcll(peers);
    __builtin_amdgcn_wave_barrier();
    if (ok) {
      pk[base + pos] = ((unsigned)(lo - rr.x) << tile_shift) | (unsigned)(c - (tstart ? tstart[t] : t * tile_cols));
      tv[base + pos] = val[k];
    }
  }
}

// ---------------------------------------------------------------- checksums (tests)

// order-sensitive 64-bit checksum of a device array of 4-byte words
__global__ __launch_bounds__(TPB) void checksum_kernel(const unsigned *__restrict__ w, int64_t n, unsigned long long *out) {
  unsigned long long acc = 0;
  const int64_t stride = (int64_t)gridDim.x * TPB;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride)
    acc += ((unsigned long long)w[i] + 0x9E3779B97F4A7C15ull) * (2ull * (unsigned long long)i + 1ull);
  atomicAdd(out, acc);
}

int device_checksum(const void *p, int64_t words, unsigned long long *host_out, hipStream_t st) {
  *host_out = 0;
  if (!p || words <= 0) return 0;
  unsigned long long *d = nullptr;
  HIP_TRY(hipMalloc((void **)&d, 8));
  HIP_TRY(hipMemsetAsync(d, 0, 8, st));
  hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(TPB), 0, st, (const unsigned *)p, words, d);
  HIP_TRY(hipMemcpyAsync(host_out, d, 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  (void)hipFree(d);
  return 0;
}

}  // namespace
