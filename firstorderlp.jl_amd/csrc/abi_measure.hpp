// abi_measure.hpp -- part of the single translation unit pdhg_hip.hip (included there, at the place its text used to stand).
// C ABI: measurement (profiling brackets, kernel names, algorithmic bytes, triad, launch overhead, the sweep-pattern probe, layout info).

// ---- measurement ------------------------------------------------------------

int pdhg_profile_enable(pdhg_handle *h, int enable) {
  // (through check_handle: switching the profile flag moves a group between the persistent launches and the per-launch path,
  //  so the members' streams must first wait for the last persistent launch)
  int rc0 = check_handle(h);
  if (rc0) return rc0;
  h->profile = enable != 0;     // a group is profiled through its first local shard
  if (enable) for (int k = 0; k < PDHG_K_COUNT; ++k) { h->prof_count[k] = 0; h->prof_ms[k] = 0.0; }
  return 0;
}

int pdhg_profile_read(pdhg_handle *h, int kernel_id, int64_t *launches, double *total_ms) {
  if (!h || kernel_id < 0 || kernel_id >= PDHG_K_COUNT) return fail(-1, "bad kernel id");
  *launches = h->prof_count[kernel_id];
  *total_ms = h->prof_ms[kernel_id];
  return 0;
}

int64_t pdhg_kernel_algorithmic_bytes(pdhg_handle *h, int kernel_id) {
  if (!h) return -1;
  // sizes of THIS shard: m rows, nnz nonzeros, cn owned columns of n
  const int64_t m = h->m, n = h->n, nnz = h->nnz, cn = h->cn;
  const bool group = h->grp != nullptr;
  switch (kernel_id) {
    // lazy accept: K7's sums are read and written where x and y are read anyway
    case PDHG_K_PRIMAL: return 8 * (7 + (h->lazy_accept ? 2 : 0)) * cn;         // r: x,c,aty,lb,ub  w: x',xbar  (+ r/w sum_x)
    case PDHG_K_SPMV_DUAL:                                                      // + r: y,b  w: y'  (+ r/w sum_y)
      return nnz * 12 + (m + 1) * 4 + n * 8 + (3 + (h->lazy_accept ? 2 : 0)) * m * 8;
    case PDHG_K_SPMV_ATY:                                                // fused: + r: x,x',aty  w: aty'
      return nnz * 12 + (n + 1) * 4 + m * 8 + (group ? 1 : 4) * n * 8;
    case PDHG_K_FINAL: return 8 * (int64_t)(3 * h->At.slots() + h->A.slots());
    case PDHG_K_ACCEPT: return 8 * 3 * (cn + m);
    case PDHG_K_ALLGATHER: return group ? 8 * (h->n_alloc - h->grp->S) : 0;        // bytes received per rank
    case PDHG_K_REDUCE_SCATTER: return group ? 8 * (h->n_alloc - h->grp->S) : 0;
    case PDHG_K_INTERACTION: return group ? 8 * 4 * cn : 0;
    default: return -1;
  }
}

namespace {
// a[i] = b[i] + s*c[i], 32 bytes per lane and pass (two 16-byte loads per stream in
// flight), one workgroup of 256 threads per 8 KiB of each stream: the access shape
// that reaches the chip's streaming rate (MI355X_MICROARCH.md: float4 copy 6.29 TB/s).
typedef double dbl2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(TPB) void triad_kernel(int64_t len4, const dbl2_t *__restrict__ b,
                                                    const dbl2_t *__restrict__ c, double s,
                                                    dbl2_t *__restrict__ a) {
  const int64_t stride = (int64_t)gridDim.x * TPB;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < len4; i += stride) {
    const int64_t k = 2 * i;
    const dbl2_t b0 = __builtin_nontemporal_load(b + k), b1 = __builtin_nontemporal_load(b + k + 1);
    const dbl2_t c0 = __builtin_nontemporal_load(c + k), c1 = __builtin_nontemporal_load(c + k + 1);
    __builtin_nontemporal_store(b0 + s * c0, a + k);
    __builtin_nontemporal_store(b1 + s * c1, a + k + 1);
  }
}
}  // namespace

int pdhg_measure_triad(pdhg_handle *h, int64_t len, int reps, double *gbps) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (len <= 0 || (len & 3) || reps <= 0 || !gbps) return fail(-1, "bad triad arguments (len must be a multiple of 4)");
  double *buf = nullptr;
  HIP_TRY(hipMalloc((void **)&buf, sizeof(double) * 3 * (size_t)len));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float best = 1e30f;
  hipError_t err = hipMemsetAsync(buf, 0, sizeof(double) * 3 * (size_t)len, h->stream);
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  const int64_t len4 = len / 4;
  const int64_t full = (len4 + TPB - 1) / TPB;                 // one pass per thread
  const int64_t grids[4] = {full, std::max<int64_t>(1, full / 2), 256 * 32, 256 * 64};
  for (int g = 0; g < 4 && err == hipSuccess; ++g) {
    const int grid = (int)std::min<int64_t>(grids[g], 1 << 30);
    for (int r = 0; r <= reps && err == hipSuccess; ++r) {   // pass 0 warms up
      (void)hipEventRecord(e0, h->stream);
      hipLaunchKernelGGL(triad_kernel, dim3(grid), dim3(TPB), 0, h->stream, len4,
                         reinterpret_cast<const dbl2_t *>(buf + len), reinterpret_cast<const dbl2_t *>(buf + 2 * len),
                         0.5, reinterpret_cast<dbl2_t *>(buf));
      (void)hipEventRecord(e1, h->stream);
      err = hipEventSynchronize(e1);
      float ms = 0.f;
      if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
      if (r > 0 && ms < best) best = ms;
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(buf);
  HIP_TRY(err);
  *gbps = 24.0 * (double)(4 * len4) / ((double)best * 1e-3) / 1e9;
  return 0;
}

namespace {
__global__ void noop_kernel(int *sink) { if (sink && threadIdx.x == 1024) *sink = 0; }
}  // namespace

extern "C++" {
namespace {
// ---- what the tiled sweep's ACCESS PATTERN can reach on this chip, with nothing else in the kernel ------------------
// The geometry of spmv_tiled_kernel -- 8-wave workgroups, two per CU (the dynamic LDS of the product is reserved, unused),
// every wave walking the same column tiles in lock step with one pacing barrier per tile, the entries of a (wave, tile)
// cell streamed as 4-byte packed offsets + 8-byte values with non-temporal loads one tile ahead, one 8-byte gather per
// entry from the tile's window of the vector -- but no accumulators, no row logic, no epilogue: the products are added
// into a register.  Its time for the same number of gathers is the floor of this design on this matrix shape; bench.py
// reports the product kernel's time against it (roofline.ceiling_frac) next to the 8 TB/s figure.  FLAT: no tiles, no
// barrier, every gather of every wave falls into ONE window of tile_cols columns -- the same wave geometry with a perfect
// cache (what tile switches and pacing cost), not the chip's all-hit rate at full occupancy.
template <bool FLAT>
__global__ __launch_bounds__(TW_WPB * WAVE) void sweep_ceiling_kernel(const unsigned *__restrict__ pk, const double *__restrict__ tv,
                                                                      const double *__restrict__ x, double *__restrict__ out,
                                                                      int nwaves, int ntiles, int tile_cols, int cnt) {
  extern __shared__ double ceiling_lds[];
  constexpr int C = 3;                                   // 64-entry chunks per cell held in registers (TW_U)
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
  const int w = blockIdx.x * TW_WPB + wid;
  const bool live = w < nwaves;
  const size_t cell = (size_t)C * WAVE;
  const unsigned *my = pk + (size_t)(live ? w : 0) * ntiles * cell;
  const double *myv = tv + (size_t)(live ? w : 0) * ntiles * cell;
  double s = 0.0;
  unsigned p[2][C];
  double v[2][C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const bool ok = live && c * WAVE + lane < cnt;
    p[0][c] = ok ? __builtin_nontemporal_load(my + c * WAVE + lane) : 0u;
    v[0][c] = ok ? __builtin_nontemporal_load(myv + c * WAVE + lane) : 0.0;
  }
  for (int t0 = 0; t0 < ntiles; t0 += 2) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int t = t0 + b;
      if (t < ntiles) {                                  // workgroup-uniform
        const double *xt = FLAT ? x : x + (size_t)t * tile_cols;
        double g[C];
#pragma unroll
        for (int c = 0; c < C; ++c) g[c] = (live && c * WAVE + lane < cnt) ? xt[p[b][c]] : 0.0;
        if (t + 1 < ntiles) {
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const bool ok = live && c * WAVE + lane < cnt;
            p[b ^ 1][c] = ok ? __builtin_nontemporal_load(my + (size_t)(t + 1) * cell + c * WAVE + lane) : 0u;
            v[b ^ 1][c] = ok ? __builtin_nontemporal_load(myv + (size_t)(t + 1) * cell + c * WAVE + lane) : 0.0;
          }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) s = s + v[b][c] * g[c];
        if (!FLAT) __syncthreads();                      // the sweep's pacing barrier
      }
    }
  }
  if (s == 0.123456789) out[0] = s + ceiling_lds[0];     // keeps the sum (and the LDS reservation) alive
}
__global__ __launch_bounds__(TPB) void ceiling_fill_kernel(unsigned *pk, double *tv, size_t len, unsigned tile_cols) {
  const size_t stride = (size_t)gridDim.x * TPB;
  for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < len; i += stride) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    pk[i] = (unsigned)(z % tile_cols);
    tv[i] = 1.0 + (double)(z >> 40) * 1e-9;
  }
}
}  // namespace
}  // extern "C++"

/* out[0]: G gathers/s of the sweep's pattern (tiles + pacing barriers + entry streams, no accumulation) for `rows` rows,
 * `cols` columns and `nnz` entries in this handle's geometry (its constraint matrix's sweep layout when it has one:
 * waves, tiles, tile width; otherwise 1221 rows per wave and the tile width the library would choose);
 * out[1]: milliseconds of one such pass; out[2]: G gathers/s when every gather falls into ONE window of the tile's width
 * and nothing synchronises (the chip's all-hit rate for 8-byte gathers beside the entry streams); out[3]: entries per
 * (wave, tile) cell; out[4] / out[5]: waves and tiles of the probe. */
int pdhg_measure_sweep_ceiling(pdhg_handle *h, int64_t rows, int64_t cols, int64_t nnz, int reps, double out[6]) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (rows <= 0 || cols <= 0 || nnz <= 0 || reps <= 0 || !out) return fail(-1, "bad ceiling-probe arguments");
  HIP_TRY(hipSetDevice(h->device));
  auto fits = [&](const CsrDev &M) { return M.tiled && M.nwaves > 0 && M.ntiles > 0 && M.rows == rows && M.cols == cols; };
  const CsrDev &D = (!fits(h->A) && fits(h->At)) ? h->At : h->A;      // the sweep layout of the product with these extents
  const bool have = fits(D);
  const int tile_cols = have ? D.tile_cols : std::max(4096, choose_tile_cols(cols, nnz, rows) > 0 ? choose_tile_cols(cols, nnz, rows) : 65536);
  const int ntiles = have ? D.ntiles : (int)((cols + tile_cols - 1) / tile_cols);
  const int tw_rows = have ? D.tw_rows : 1221;
  const int nwaves = have ? D.nwaves : (int)((rows + tw_rows - 1) / tw_rows);
  const int cnt = (int)std::min<int64_t>(3 * WAVE, std::max<int64_t>(1, (nnz + (int64_t)nwaves * ntiles / 2) / ((int64_t)nwaves * ntiles)));
  const size_t len = (size_t)nwaves * ntiles * 3 * WAVE;
  if (len > ((size_t)1 << 32)) return fail(-2, "ceiling probe: geometry too large");
  unsigned *pk = nullptr;
  double *tv = nullptr, *x = nullptr, *o = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t err = hipMalloc((void **)&pk, sizeof(unsigned) * len);
  if (err == hipSuccess) err = hipMalloc((void **)&tv, sizeof(double) * len);
  if (err == hipSuccess) err = hipMalloc((void **)&x, sizeof(double) * ((size_t)ntiles * tile_cols + 16));
  if (err == hipSuccess) err = hipMalloc((void **)&o, 64);
  if (err == hipSuccess) err = hipMemsetAsync(x, 0, sizeof(double) * ((size_t)ntiles * tile_cols + 16), h->stream);
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  double best[2] = {1e30, 1e30};
  if (err == hipSuccess) {
    hipLaunchKernelGGL(ceiling_fill_kernel, dim3(4096), dim3(TPB), 0, h->stream, pk, tv, len, (unsigned)tile_cols);
    const size_t lds = have ? tiled_lds_bytes(D) : (size_t)78 * 1024;     // two workgroups per CU, as the product runs
    err = hipFuncSetAttribute((const void *)sweep_ceiling_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err == hipSuccess) err = hipFuncSetAttribute((const void *)sweep_ceiling_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = (nwaves + TW_WPB - 1) / TW_WPB;
    for (int flat = 0; flat < 2 && err == hipSuccess; ++flat) {
      for (int r = 0; r <= reps && err == hipSuccess; ++r) {               // pass 0 warms up
        (void)hipEventRecord(e0, h->stream);
        if (flat) hipLaunchKernelGGL(sweep_ceiling_kernel<true>, dim3(grid), dim3(TW_WPB * WAVE), lds, h->stream, pk, tv, x, o, nwaves, ntiles, tile_cols, cnt);
        else hipLaunchKernelGGL(sweep_ceiling_kernel<false>, dim3(grid), dim3(TW_WPB * WAVE), lds, h->stream, pk, tv, x, o, nwaves, ntiles, tile_cols, cnt);
        (void)hipEventRecord(e1, h->stream);
        err = hipEventSynchronize(e1);
        float ms = 0.f;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best[flat]) best[flat] = ms;
      }
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  for (void *q : {(void *)pk, (void *)tv, (void *)x, (void *)o}) if (q) (void)hipFree(q);
  HIP_TRY(err);
  const double gathers = (double)nwaves * ntiles * cnt;
  out[0] = gathers / (best[0] * 1e-3) / 1e9;
  out[1] = best[0];
  out[2] = gathers / (best[1] * 1e-3) / 1e9;
  out[3] = (double)cnt;
  out[4] = (double)nwaves;
  out[5] = (double)ntiles;
  return 0;
}

/* Phase timeline of the last one-launch trial (needs PDHG_COOP_TRACE=1 in the environment when the handle takes its
 * first one-launch trial): see trial_timeline.  Returns 1 when no trace was recorded.  Measurement only. */
int pdhg_trial_timeline(pdhg_handle *h, double out[14]) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!out) return fail(-1, "out == NULL");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return trial_timeline(h, out) ? fail(1, "no one-launch trial has been traced (PDHG_COOP_TRACE=1, stream-layout LP on one handle)") : 0;
}

int pdhg_selftest_wave_sums(pdhg_handle *h, int64_t seed, int64_t out[2]) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!out) return fail(-1, "out == NULL");
  unsigned long long *bad = nullptr;
  HIP_TRY(hipMalloc((void **)&bad, sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(bad, 0, sizeof(unsigned long long), h->stream));
  constexpr int GRID = 256;
  hipLaunchKernelGGL(wave_sums_selftest_kernel<8>, dim3(GRID), dim3(TPB), 0, h->stream, (unsigned long long)seed, bad);
  hipLaunchKernelGGL(wave_sums_selftest_kernel<TR_SETUP_NS>, dim3(GRID), dim3(TPB), 0, h->stream, (unsigned long long)seed + 1, bad);
  hipLaunchKernelGGL(wave_sums_selftest_kernel<22>, dim3(GRID), dim3(TPB), 0, h->stream, (unsigned long long)seed + 2, bad);
  hipLaunchKernelGGL((wave_sums_selftest_kernel<TR_Q * TR_K>), dim3(GRID), dim3(TPB), 0, h->stream, (unsigned long long)seed + 3, bad);
  hipLaunchKernelGGL(wave_sums_selftest_kernel<64>, dim3(GRID), dim3(TPB), 0, h->stream, (unsigned long long)seed + 4, bad);
  unsigned long long host_bad = 0;
  hipError_t e = hipMemcpyAsync(&host_bad, bad, sizeof host_bad, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(bad);
  if (e != hipSuccess) return fail_hip(e, "pdhg_selftest_wave_sums");
  out[0] = (int64_t)GRID * (TPB / WAVE) * (8 + TR_SETUP_NS + 22 + TR_Q * TR_K + 64);
  out[1] = (int64_t)host_bad;
  return 0;
}

int pdhg_measure_launch_overhead(pdhg_handle *h, int reps, double out[2]) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (reps <= 0 || !out) return fail(-1, "bad arguments");
  HIP_TRY(hipStreamSynchronize(h->stream));
  double best[2] = {1e30, 1e30};
  for (int k = 1; k <= 2; ++k)
    for (int r = 0; r <= reps; ++r) {          // pass 0 warms up
      HIP_TRY(hipEventRecord(h->ev0, h->stream));
      for (int q = 0; q < (k == 1 ? 1 : 5); ++q) hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, h->stream, (int *)nullptr);
      HIP_TRY(hipEventRecord(h->ev1, h->stream));
      HIP_TRY(hipEventSynchronize(h->ev1));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, h->ev0, h->ev1));
      if (r > 0 && ms < best[k - 1]) best[k - 1] = ms;
    }
  out[0] = best[0];                              // one empty launch between two events
  out[1] = (best[1] - best[0]) / 4.0;            // every further launch inside the same bracket
  return 0;
}

int pdhg_layout_checksums(pdhg_handle *h, uint64_t out[32]) {
  int rc = check_handle(h);
  if (rc) return rc;
  if (!out) return fail(-1, "out == NULL");
  if (!h->A.segs.empty() || !h->At.segs.empty()) return fail(-2, "layout checksums are per piece: not defined for a matrix held as row segments");
  HIP_TRY(hipStreamSynchronize(h->stream));
  const CsrDev *Ls[2] = {&h->A, &h->At};
  for (int k = 0; k < 2; ++k) {
    const CsrDev &D = *Ls[k];
    const struct { const void *p; int64_t words; } parts[16] = {
        {D.rowptr, (int64_t)D.rows + 1}, {D.col, D.nnz}, {D.val, 2 * D.nnz}, {D.blks, 2 * (int64_t)D.nblk},
        {D.long_row, D.nlong}, {D.long_chunk_ptr, (int64_t)D.nlong + 1}, {D.chunk_row, D.nchunks}, {D.chunk_off, D.nchunks},
        {D.tiled ? D.pk : nullptr, D.tw_entries}, {D.tiled ? D.tv : nullptr, 2 * D.tw_entries},
        {D.tiled ? D.wave_rows : nullptr, 2 * (int64_t)D.nwaves}, {D.tiled ? D.wave_ent : nullptr, D.step_ptr_len},
        {D.tiled ? D.wave_step_off : nullptr, D.nwaves}, {D.tiled ? D.step_tile : nullptr, D.total_steps},
        {D.tiled ? D.wg_step_off : nullptr, D.tiled ? (int64_t)D.grid + 1 : 0}, {nullptr, 0}};
    for (int q = 0; q < 16; ++q) {
      unsigned long long v = 0;
      if ((rc = device_checksum(parts[q].p, parts[q].words, &v, h->stream))) return rc;
      out[16 * k + q] = v;
    }
    // a stream layout's column slabs ride in the sweep's (then unused) slots: row pointers, columns, values, row blocks
    if (!D.tiled) {
      for (size_t s = 0; s < D.slabs.size(); ++s) {
        const SlabDev &S = D.slabs[s];
        const struct { const void *p; int64_t words; } sp4[4] = {
            {S.rowptr, (int64_t)D.rows + 1}, {S.col, S.nnz}, {S.val, 2 * S.nnz}, {S.blks, 2 * (int64_t)S.nblk}};
        for (int q = 0; q < 4; ++q) {
          unsigned long long v = 0;
          if ((rc = device_checksum(sp4[q].p, sp4[q].words, &v, h->stream))) return rc;
          out[16 * k + 8 + q] += v * (2ull * s + 3ull);
        }
      }
    }
    // the plan's scalars ride in the last slot
    out[16 * k + 15] = (uint64_t)D.tiled + 2ull * (uint64_t)std::min(D.tw_mode, 3) + 8ull * (uint64_t)D.tile_shift + 1024ull * (uint64_t)D.tw_rows +
                       (1ull << 32) * (uint64_t)D.grid;
  }
  return 0;
}

// One matrix's resident layout as a JSON object (pdhg_layout_describe).
static std::string describe_matrix(const pdhg_handle *h, const CsrDev &D0, int mode, int tag) {
  const CsrDev &D = D0.segs.empty() ? D0 : D0.segs.front();
  char b[512];
  std::string out = "{";
  snprintf(b, sizeof b, "\"rows\": %d, \"cols\": %d, \"nnz\": %lld, \"segments\": %d, \"kernels\": \"%s\"", D0.rows, D0.cols, (long long)D0.nnz,
           (int)D0.segs.size(), product_kernels(D0, mode, tag).c_str());
  out += b;
  const char *kind = D.tiled ? "sweep" : ((D.sj.on() || (!D.slabs.empty() && D.slabs.front().sj.on())) ? "sliced jagged"
                     : ((D.pipe_grid > 0 || (!D.slabs.empty() && D.slabs.front().pipe_grid > 0)) ? "row blocks, pipelined" : "row blocks"));
  snprintf(b, sizeof b, ", \"layout\": \"%s\", \"row_blocks\": %d, \"long_rows\": %d, \"long_threshold\": %d, \"column_slabs\": %d, \"max_row_nnz\": %lld",
           kind, D.nblk, D.nlong, D.long_thr, (int)D.slabs.size(), (long long)D0.max_row_nnz);
  out += b;
  if (D.tiled) {
    snprintf(b, sizeof b, ", \"sweep\": {\"waves\": %d, \"tile_cols\": %d, \"equal_nonzero_tiles\": %s, \"chunk_variant\": %d, \"xcd_dealing\": \"%s\", "
             "\"chosen_by\": \"%s\", \"candidates\": [", D.nwaves, D.tile_cols, D.var_tiles ? "true" : "false", D.tw_mode,
             tiled_per_xcd(h, D, D.grid) > 0 ? "contiguous eighths" : "round robin", D.tw_tuned > 1 ? "timing at create (tune_tiled_variant)" : "static rule");
    out += b;
    for (int k = 0; k < D.tw_tuned; ++k) {
      snprintf(b, sizeof b, "%s{\"chunk_variant\": %d, \"xcd_dealing\": \"%s\", \"ms\": %.4f}", k ? ", " : "", D.tw_tune_mode[k],
               D.tw_tune_band[k] ? "contiguous eighths" : "round robin", (double)D.tw_tune_ms[k]);
      out += b;
    }
    out += "]}";
  }
  const SjDev *J = D.sj.on() ? &D.sj : ((!D.slabs.empty() && D.slabs.front().sj.on()) ? &D.slabs.front().sj : nullptr);
  if (J) {
    int64_t hub = 0, hub_nnz = 0;
    if (D.sj.on()) { hub = D.sj.nhub; hub_nnz = D.sj.hub_nnz; }
    else for (const SlabDev &S : D.slabs) { hub += S.sj.nhub; hub_nnz += S.sj.hub_nnz; }
    snprintf(b, sizeof b, ", \"sliced_jagged\": {\"window_rows\": %d, \"slices_per_wave\": %d, \"hub_threshold\": %d, \"hub_rows\": %lld, \"hub_nnz\": %lld, "
             "\"fill_narrow\": %.3f, \"fill_wide\": %.3f, \"ragged_share\": %.3f, \"grid\": %d}", SJ_SIGMA * J->G, J->G, J->max_len, (long long)hub, (long long)hub_nnz,
             J->fill_narrow, J->fill_wide, J->ragged, J->grid);
    out += b;
  }
  return out + "}";
}

// The resident layouts and every choice pdhg_create made for them -- including the ones settled by TIMING the product on the
// matrix (the sweep's chunk variant and XCD dealing: tune_tiled_variant) -- as one JSON object, so that a bench line or a
// profile can say which variant it ran (a handle's dispatch depends on a measurement taken at create; PDHG_TUNE=0 pins the
// static rules).  Returns the length of the text (without the terminator); writes at most cap - 1 characters + NUL.
int pdhg_layout_describe(pdhg_handle *h, char *buf, int cap) {
  if (!h) return fail(-1, "null handle");
  std::string out = "{\"A\": " + describe_matrix(h, h->A, MODE_DUAL, 0) + ", \"At\": " + describe_matrix(h, h->At, h->grp ? MODE_PLAIN : MODE_ATY, 1);
  if (h->has_q) out += ", \"Q\": " + describe_matrix(h, h->Q, MODE_PLAIN, 2) + ", \"Qt\": " + describe_matrix(h, h->Qt, MODE_PLAIN, 2);
  if (h->grp && !h->Achunk.empty()) {
    // the all-gather of xbar cut into column chunks, A_p xbar as one pass per chunk (dist.hpp: DistGroup::ag_chunks)
    char b[256];
    snprintf(b, sizeof b, ", \"all_gather\": {\"chunks\": %d, \"mode\": \"%s\", \"columns_per_rank_and_chunk\": %lld, \"passes\": [",
             (int)h->Achunk.size(), (h->grp->ag_mode == 1 && h->grp->backend == COMM_RCCL) ? "overlapped with A_p xbar" : "passes behind one all-gather",
             (long long)h->grp->ag_sub);
    out += b;
    for (size_t c = 0; c < h->Achunk.size(); ++c)
      out += (c ? ", " : "") + describe_matrix(h, h->Achunk[c], c + 1 < h->Achunk.size() ? MODE_PLAIN : MODE_DUAL, 0);
    out += "]}";
  }
  const char *tv = getenv("PDHG_TUNE");
  out += std::string(", \"row_order\": \"") + (h->relaxed ? "relaxed" : "strict") + "\", \"timing_at_create\": " + ((tv && tv[0] == '0') ? "false" : "true") + "}";
  if (buf && cap > 0) {
    const size_t k = std::min(out.size(), (size_t)cap - 1);
    memcpy(buf, out.data(), k);
    buf[k] = 0;
  }
  return (int)out.size();
}

int pdhg_layout_info(pdhg_handle *h, int64_t info[16]) {
  if (!h) return fail(-1, "null handle");
  info[12] = (int64_t)h->A.slabs.size(); info[13] = (int64_t)h->At.slabs.size();
  // 2: one persistent kernel per trial (trial_kernel.hpp), 1: one graph launch, 0: separate launches
  info[14] = (coop_eligible(h) || h->coop_mode == 1) ? 2 : ((graph_eligible(h) || (h->graph_mode == 1 && !h->has_q)) ? 1 : 0);
  info[15] = ((h->A.tiled && h->A.var_tiles) || (!h->A.segs.empty() && h->A.segs.front().tiled && h->A.segs.front().var_tiles) ? 1 : 0) +
             ((h->At.tiled && h->At.var_tiles) || (!h->At.segs.empty() && h->At.segs.front().tiled && h->At.segs.front().var_tiles) ? 2 : 0) +
             (small_lp_eligible(h) ? 4 : 0) +
             (!h->grp && !h->has_q && !small_lp_eligible(h) && device_loop_for(h) && coop_eligible(h) ? 8 : 0) +
             (h->local_mode == 1 && h->local_launches > 0 ? 16 : 0);      // the multi-step kernel runs in its XCD-local mode
  // a matrix held as row segments (64-bit extents, layout.hpp) reports the sums over its segments, the first segment's
  // tile width, and the segment counts in bits 8-15 (A) and 16-23 (A') of info[15]
  auto total = [](const CsrDev &D, auto f) { int64_t t = 0; if (D.segs.empty()) return (int64_t)f(D); for (const CsrDev &S : D.segs) t += f(S); return t; };
  auto first = [](const CsrDev &D) -> const CsrDev & { return D.segs.empty() ? D : D.segs.front(); };
  const CsrDev *Ms[2] = {&h->A, &h->At};
  for (int k = 0; k < 2; ++k) {
    const CsrDev &D = *Ms[k];
    info[4 * k + 0] = total(D, [](const CsrDev &S) { return S.nblk; });
    info[4 * k + 1] = total(D, [](const CsrDev &S) { return S.nlong; });
    info[4 * k + 2] = total(D, [](const CsrDev &S) { return S.nchunks; });
    info[4 * k + 3] = D.max_row_nnz;
    info[8 + k] = total(D, [](const CsrDev &S) { return S.tiled ? S.nwaves : 0; });
    info[10 + k] = first(D).tiled ? first(D).tile_cols : 0;
    info[15] += (int64_t)std::min<size_t>(D.segs.size(), 255) << (8 + 8 * k);
  }
  if (h->grp)     // bits 24-39: trials this group took as one persistent kernel per shard (group_kernel.hpp); 40-47: its fallbacks
    info[15] += (std::min<int64_t>(h->grp->coop_trials, 65535) << 24) + ((int64_t)std::min(h->grp->coop_fallbacks, 255) << 40);
  info[15] += std::min<int64_t>(h->tr_coop_calls, 16383) << 48;      // trust-region calls taken as one persistent launch
  if (!h->A.segs.empty()) info[12] = (int64_t)first(h->A).slabs.size();
  if (!h->At.segs.empty()) info[13] = (int64_t)first(h->At).slabs.size();
  // bit 8 of the slab counts: the product runs on the sliced jagged layout (sj_kernels.hpp)
  for (int k = 0; k < 2; ++k) {
    const CsrDev &D = first(*Ms[k]);
    if (D.sj.on() || (!D.slabs.empty() && D.slabs.front().sj.on())) info[12 + k] += 256;
    if (D.pipe_grid > 0 || (!D.slabs.empty() && D.slabs.front().pipe_grid > 0)) info[12 + k] += 512;   // bit 9: spmv_stream_pipe_kernel
  }
  return 0;
}

