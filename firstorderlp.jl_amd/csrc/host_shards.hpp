// host_shards.hpp -- part of the single translation unit pdhg_hip.hip (included there, at the place its text used to stand).
// moving distributed vectors; creating a shard, the segments of a matrix beyond 2^31 nonzeros, a shard group (host side).

// ---- moving distributed vectors -------------------------------------------------
// "column vectors": n-vectors whose valid part on a shard is its own slice;
// "row vectors": m-vectors, each shard holds its rows.

// full copy of a column vector on every shard's device: dst[0..n) (dst holds n_alloc)
template <typename Src, typename Dst>
int gather_cols_device(const Shards &L, Src src, Dst dst) {
  if (!L.g) {
    pdhg_handle *s = L.p[0];
    if (src(s) != dst(s))
      HIP_TRY(hipMemcpyAsync(dst(s), src(s), sizeof(double) * (size_t)s->n, hipMemcpyDeviceToDevice, s->stream));
    return 0;
  }
  FOR_SHARDS(L, s) {
    if (src(s) != dst(s) && s->cn > 0)
      HIP_TRY(hipMemcpyAsync(dst(s) + s->clo, src(s) + s->clo, sizeof(double) * (size_t)s->cn,
                             hipMemcpyDeviceToDevice, s->stream));
  }
  return dist_all_gather(*L.g, dst, L.g->S);
}

// a column vector to a host array of length n (every process gets all of it); caller syncs
template <typename Src>
int cols_to_host(const Shards &L, Src src, double *host) {
  if (!L.g || L.g->all_local()) {
    FOR_SHARDS(L, s) {
      if (s->cn > 0)
        HIP_TRY(hipMemcpyAsync(host + s->clo, src(s) + s->clo, sizeof(double) * (size_t)s->cn, hipMemcpyDeviceToHost, s->stream));
    }
    return 0;
  }
  int rc = gather_cols_device(L, src, [](pdhg_handle *s) { return s->dn_buf; });
  if (rc) return rc;
  pdhg_handle *s = L.p[0];
  HIP_TRY(hipMemcpyAsync(host, s->dn_buf, sizeof(double) * (size_t)s->n, hipMemcpyDeviceToHost, s->stream));
  return 0;
}

// a row vector to a host array of length m_global; caller syncs
template <typename Src>
int rows_to_host(const Shards &L, Src src, double *host) {
  if (!L.g || L.g->all_local()) {
    FOR_SHARDS(L, s) {
      if (s->m > 0)
        HIP_TRY(hipMemcpyAsync(host + s->row_lo, src(s), sizeof(double) * (size_t)s->m, hipMemcpyDeviceToHost, s->stream));
    }
    return 0;
  }
  pdhg_handle *s = L.p[0];
  HIP_TRY(hipSetDevice(s->device));
  if (s->m > 0)
    HIP_TRY(hipMemcpyAsync(s->dm_buf + s->row_lo, src(s), sizeof(double) * (size_t)s->m, hipMemcpyDeviceToDevice, s->stream));
  int rc = dist_all_gather_rows(*L.g, [](pdhg_handle *q) { return q->dm_buf; });
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(host, s->dm_buf, sizeof(double) * (size_t)s->m_global, hipMemcpyDeviceToHost, s->stream));
  return 0;
}

// host arrays (global length) to the shards: column vectors are stored in full, row vectors by rows
template <typename Dst>
int cols_from_host(const Shards &L, const double *host, Dst dst) {
  FOR_SHARDS(L, s) {
    if (s->n > 0) HIP_TRY(hipMemcpyAsync(dst(s), host, sizeof(double) * (size_t)s->n, hipMemcpyHostToDevice, s->stream));
  }
  return 0;
}
template <typename Dst>
int rows_from_host(const Shards &L, const double *host, Dst dst) {
  FOR_SHARDS(L, s) {
    if (s->m > 0)
      HIP_TRY(hipMemcpyAsync(dst(s), host + s->row_lo, sizeof(double) * (size_t)s->m, hipMemcpyHostToDevice, s->stream));
  }
  return 0;
}

// A'y for a row vector y (each shard its rows) into the column vector `out`
// (valid on the owned slice; out holds n_alloc in a group).
template <typename Yin, typename Out>
int dual_product(const Shards &L, Yin yin, Out out) {
  int rc;
  FOR_SHARDS(L, s) { if ((rc = launch_aty_plain(s, yin(s), out(s)))) return rc; }
  if (L.g) {
    ProfScope ps(L.p[0], PDHG_K_REDUCE_SCATTER);
    if ((rc = dist_reduce_scatter(*L.g, out, L.g->S))) return rc;
  }
  return 0;
}

// Layout choice for one CSR: the tiled sweep pays off when the gathered vector
// (cols doubles) is comparable to or larger than an XCD's 4 MiB L2.
// PDHG_SPMV=stream|tiled forces a layout; PDHG_TILE_SHIFT sets log2(tile cols).
int choose_tile_cols(int64_t cols, int64_t nnz, int64_t rows) {
  const char *mode = getenv("PDHG_SPMV");
  const char *ts = dev_env("PDHG_TILE_SHIFT"), *tc = getenv("PDHG_TILE_COLS");
  int64_t tile = tc ? atoll(tc) : (ts ? (1LL << std::min(22, std::max(6, atoi(ts)))) : 65536);
  bool thin = false;
  if (!ts && !tc && rows > 0) {
    // One step of the sweep costs about the same for any cell of up to TW_U x 64
    // entries, and a smaller tile keeps the gathered vector in L2 more reliably, so
    // the tile is as narrow as a cell of ~100-110 entries allows (entries per wave /
    // number of tiles; tile widths are multiples of 4096 columns, not powers of two).
    // Measured on MI355X (profiles/r02_tile_rule.txt), time per nonzero against
    // entries per cell on the same matrix: 40 -> +20 %, 48-64 -> +10-18 %,
    // 88-120 -> best, 160 -> +15-20 %.  The width is capped where the L2 stops
    // holding the tile against the entry stream: 76K columns (608 KiB) when several
    // residency rounds are in flight (config S: 0.73 ms at 72-80K, 0.96 ms at 96K;
    // 16M x 16M: 1.33 ms at 80K, 1.58 ms at 96K), 144K columns for a single round
    // (row shards of a multi-GPU run).
    const int64_t slots = 256LL * 2 * TW_WPB;
    const int64_t rounds = std::max<int64_t>(1, (rows + slots * TW_MAX_ROWS - 1) / (slots * TW_MAX_ROWS));
    const int64_t rpw = std::max<int64_t>(64, (rows + slots * rounds - 1) / (slots * rounds));
    const double per_wave = (double)nnz / (double)rows * (double)rpw;
    const double target = dev_env("PDHG_TILE_FILL") ? atof(dev_env("PDHG_TILE_FILL")) : (rounds == 1 ? 110.0 : 100.0);
    // Many rounds (>= 5, i.e. beyond ~21M rows): the cells thin out at the 76K cap and the balance tips
    // towards wider tiles -- 24M x 24M 2.51 ms at 76K columns / 2.39 at 96K, 30M x 30M 3.47 / 3.08 / 2.92 at
    // 76K / 96K / 112K (128K: 3.46), 20M and below indifferent or worse -- so the cap stretches to what
    // gives a cell ~45 entries, up to 112K.
    const int64_t unit = 4096;
    int64_t cap = tile_width_cap(rows);
    if (rounds >= 2) {
      const int64_t stretch = ((int64_t)((double)cols * 45.0 / std::max(per_wave, 1.0)) + unit - 1) / unit * unit;
      cap = std::max(cap, std::min<int64_t>(112 * 1024, stretch));
    }
    const int64_t want = (int64_t)((double)cols * target / std::max(per_wave, 1.0));
    tile = std::min(cap, std::max<int64_t>(2 * unit, (want + unit / 2) / unit * unit));
    // Few rows against a very long vector: even the widest tile leaves a wave a handful of
    // entries per step and the sweep is all barriers (100K x 10M, 9 per cell: 0.081 ms swept,
    // 0.029 ms streamed; 1M x 30M, 12 per cell: 0.248 / 0.220; 500K x 10M, 18 per cell: 0.099 / 0.109).
    thin = per_wave * (double)tile / (double)std::max<int64_t>(cols, 1) < 15.0;
  }
  tile = std::min<int64_t>(std::max<int64_t>(tile, 64), 1LL << 22);   // leave >= 10 bits for row_local
  if ((cols + tile - 1) / tile > 65536) return 0;        // tile table would be huge
  if (mode && !strcmp(mode, "stream")) return 0;
  if (mode && !strcmp(mode, "tiled")) return (int)tile;
  // The sweep is tried whenever the gathered vector is beyond ~3 MiB; build_tiled() then
  // declines for matrices it does not suit (a row with a long run inside one tile, rows
  // that stay inside a band of columns) and the stream layout takes over.  Measured
  // crossover on square 10-per-row LPs, whole iterations: 250K columns stream 13.9k / swept
  // 13.3k it/s (the stream layout's trial is one graph launch), 500K columns 8.3k / 8.9k,
  // 1M 4.6k / 5.6k.  Row length is no criterion: 100 per row (10M x 1M transposed) streams
  // at 1.91 ms and sweeps at 0.62 ms; 1 000 per row 1.85 / 1.33 ms (profiles/r02_locality.txt).
  const bool big_vector = cols * 8 > (3LL << 20);
  return (big_vector && rows > 0 && !thin) ? (int)tile : 0;
}

int host_threads() {
  int threads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char *ev = getenv("PDHG_HOST_THREADS")) threads = std::max(1, atoi(ev));
  return threads;
}

template <typename F>
void run_threads(int T, F f) {
  if (T <= 1) { f(0); return; }
  std::vector<std::thread> pool;
  for (int t = 0; t < T; ++t) pool.emplace_back([=, &f] { f(t); });
  for (std::thread &th : pool) th.join();
}

// CSC (any int64 base) -> int32 CSR of the transpose (direct) and CSR (stable sort by row).
// The sort is a two-level bucket sort on host threads, O(nnz) work in total:
// thread t scans ITS column range and appends every entry to the bucket of the
// entry's row range (T buckets; per-(thread, bucket) output segments come from a
// small T x T count table, so bucket b holds its entries in ascending column
// order); thread b then counting-sorts bucket b by row.  Each row receives its
// entries in ascending column order -- what the sequential loop produces -- and
// the result does not depend on T.
int csc_to_both(int64_t rows, int64_t cols, int64_t nnz, const int64_t *colptr,
                const int64_t *rowval, const double *nzval, int base,
                std::vector<int> &t_rowptr, ivec &t_col, dvec &t_val,
                std::vector<int> &rowptr, ivec &col, dvec &val) {
  if (rows < 0 || cols < 0 || nnz < 0) return fail(-1, "negative dimension");
  if (rows >= INT32_MAX || cols >= INT32_MAX || nnz >= INT32_MAX || rows + cols >= INT32_MAX)
    return fail(-2, "m, n, m + n or nnz >= 2^31 need the 64-bit index path (not built)");
  if (colptr[0] != base) return fail(-1, "colptr[0] != index_base");
  if (colptr[cols] - base != nnz) return fail(-1, "colptr[n] - base != nnz");
  t_rowptr.resize(cols + 1);
  for (int64_t j = 0; j <= cols; ++j) {
    const int64_t v = colptr[j] - base;
    if (v < 0 || v > nnz || (j > 0 && v < t_rowptr[j - 1])) return fail(-1, "colptr not monotone");
    t_rowptr[j] = (int)v;
  }
  t_col.resize(nnz);
  t_val.resize(nnz);
  rowptr.assign(rows + 1, 0);
  col.resize(nnz);
  val.resize(nnz);
  const int T = (nnz >= (1 << 22) && rows >= 1024 && cols >= 1024) ? host_threads() : 1;
  std::atomic<int> bad{0};
  if (T == 1) {
    for (int64_t k = 0; k < nnz; ++k) {
      const int64_t r = rowval[k] - base;
      if (r < 0 || r >= rows) return fail(-1, "rowval out of range");
      t_col[k] = (int)r;
      t_val[k] = nzval[k];
      rowptr[r + 1] += 1;
    }
    for (int64_t i = 0; i < rows; ++i) rowptr[i + 1] += rowptr[i];
    std::vector<int> next(rowptr.begin(), rowptr.end() - 1);
    for (int64_t j = 0; j < cols; ++j)
      for (int k = t_rowptr[j]; k < t_rowptr[j + 1]; ++k) {
        const int p = next[t_col[k]]++;
        col[p] = (int)j;
        val[p] = t_val[k];
      }
    return 0;
  }
  const int64_t rpb = (rows + T - 1) / T;                 // rows per bucket
  auto col_begin = [&](int t) { return (int64_t)cols * t / T; };
  std::vector<int64_t> cnt((size_t)T * T, 0);             // cnt[t*T + b]
  // pass 1: CSR(A') = the CSC input narrowed to 32 bits; bucket counts
  run_threads(T, [&](int t) {
    int64_t *c = cnt.data() + (size_t)t * T;
    for (int64_t k = t_rowptr[col_begin(t)]; k < t_rowptr[col_begin(t + 1)]; ++k) {
      const int64_t r = rowval[k] - base;
      if (r < 0 || r >= rows) { bad.store(1); return; }
      t_col[k] = (int)r;
      t_val[k] = nzval[k];
      c[r / rpb] += 1;
    }
  });
  if (bad.load()) return fail(-1, "rowval out of range");
  std::vector<int64_t> off((size_t)T * T), bstart((size_t)T + 1, 0);
  for (int b = 0; b < T; ++b) {
    int64_t run = bstart[b];
    for (int t = 0; t < T; ++t) { off[(size_t)t * T + b] = run; run += cnt[(size_t)t * T + b]; }
    bstart[b + 1] = run;
  }
  // pass 2: scatter (row, col, val) into the buckets
  ivec brow((size_t)nnz), bcol((size_t)nnz);
  dvec bval((size_t)nnz);
  run_threads(T, [&](int t) {
    int64_t *o = off.data() + (size_t)t * T;
    for (int64_t j = col_begin(t); j < col_begin(t + 1); ++j)
      for (int k = t_rowptr[j]; k < t_rowptr[j + 1]; ++k) {
        const int r = t_col[k];
        const int64_t p = o[r / rpb]++;
        brow[p] = r; bcol[p] = (int)j; bval[p] = t_val[k];
      }
  });
  // pass 3: row counts (every bucket owns its rows), serial prefix, placement
  run_threads(T, [&](int b) {
    for (int64_t p = bstart[b]; p < bstart[b + 1]; ++p) rowptr[brow[p] + 1] += 1;
  });
  for (int64_t i = 0; i < rows; ++i) rowptr[i + 1] += rowptr[i];
  run_threads(T, [&](int b) {
    const int64_t r0 = std::min<int64_t>(rows, rpb * b), r1 = std::min<int64_t>(rows, rpb * (b + 1));
    std::vector<int> next(rowptr.begin() + r0, rowptr.begin() + r1);
    for (int64_t p = bstart[b]; p < bstart[b + 1]; ++p) {
      const int q = next[brow[p] - r0]++;
      col[q] = bcol[p];
      val[q] = bval[p];
    }
  });
  return 0;
}

// Large matrices: upload the caller's CSC arrays as they are and build CSR(A'), CSR(A) in HBM
// (device_layout.hpp).  On return A / At hold rowptr, col, val on the device and the two host vectors the
// row pointers (the only per-row data the host-side planning needs).  Validation as csc_to_both's.
int ingest_on_device(CsrDev &A, CsrDev &At, int64_t rows, int64_t cols, int64_t nnz, const int64_t *colptr,
                     const int64_t *rowval, const double *nzval, int base, std::vector<int> &rowptr,
                     std::vector<int> &t_rowptr) {
  if (rows >= INT32_MAX || cols >= INT32_MAX || nnz >= INT32_MAX || rows + cols >= INT32_MAX)
    return fail(-2, "m, n, m + n or nnz >= 2^31 need the 64-bit index path (not built)");
  if (colptr[0] != base) return fail(-1, "colptr[0] != index_base");
  if (colptr[cols] - base != nnz) return fail(-1, "colptr[n] - base != nnz");
  hipStream_t st = nullptr;
  int64_t *d_colptr = nullptr, *d_rowval = nullptr;
  int *key = nullptr, *key2 = nullptr, *col2 = nullptr, *row_cnt = nullptr, *flags = nullptr;
  double *val2 = nullptr;
  auto cleanup = [&]() {
    for (void *p : {(void *)d_colptr, (void *)d_rowval, (void *)key, (void *)key2, (void *)col2, (void *)row_cnt, (void *)flags, (void *)val2})
      if (p) (void)hipFree(p);
  };
#define DL(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { cleanup(); return fail_hip(_e, #expr); } } while (0)
  const size_t nz = (size_t)std::max<int64_t>(nnz, 1);
  DL(hipMalloc((void **)&d_colptr, sizeof(int64_t) * (size_t)(cols + 1)));
  DL(hipMalloc((void **)&d_rowval, sizeof(int64_t) * nz));
  DL(hipMalloc((void **)&At.rowptr, sizeof(int) * (size_t)(cols + 1)));
  DL(hipMalloc((void **)&At.col, sizeof(int) * nz));
  DL(hipMalloc((void **)&At.val, sizeof(double) * nz));
  DL(hipMalloc((void **)&A.rowptr, sizeof(int) * (size_t)(rows + 1)));
  DL(hipMalloc((void **)&A.col, sizeof(int) * nz));
  DL(hipMalloc((void **)&A.val, sizeof(double) * nz));
  DL(hipMalloc((void **)&key, sizeof(int) * nz));
  DL(hipMalloc((void **)&key2, sizeof(int) * nz));
  DL(hipMalloc((void **)&col2, sizeof(int) * nz));
  DL(hipMalloc((void **)&val2, sizeof(double) * nz));
  DL(hipMalloc((void **)&row_cnt, sizeof(int) * (size_t)(rows + 1)));
  DL(hipMalloc((void **)&flags, sizeof(int)));
  DL(hipMemcpy(d_colptr, colptr, sizeof(int64_t) * (size_t)(cols + 1), hipMemcpyHostToDevice));
  DL(hipMemcpy(d_rowval, rowval, sizeof(int64_t) * (size_t)nnz, hipMemcpyHostToDevice));
  DL(hipMemcpy(At.val, nzval, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice));
  DL(hipMemsetAsync(row_cnt, 0, sizeof(int) * (size_t)(rows + 1), st));
  DL(hipMemsetAsync(flags, 0, sizeof(int), st));
  hipLaunchKernelGGL(ingest_colptr_kernel, dim3((unsigned)((cols + 1 + TPB - 1) / TPB)), dim3(TPB), 0, st,
                     (const int64_t *)d_colptr, cols, nnz, base, At.rowptr, flags);
  hipLaunchKernelGGL(ingest_entries_kernel, dim3(4096), dim3(TPB), 0, st, (const int64_t *)d_rowval, (const int *)At.rowptr, nnz,
                     rows, cols, base, At.col, key, A.col, row_cnt, flags);
  int hflags = 0;
  DL(hipMemcpy(&hflags, flags, sizeof(int), hipMemcpyDeviceToHost));
  if (hflags) { cleanup(); return fail(-1, (hflags & 2) ? "colptr not monotone" : "rowval out of range"); }
  int rc = device_exclusive_scan(row_cnt, A.rowptr, rows + 1, nullptr, st);
  if (rc) { cleanup(); return rc; }
  DL(hipMemcpyAsync(A.val, At.val, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToDevice, st));
  int key_bits = 1;
  while ((1LL << key_bits) < rows) ++key_bits;
  bool in_scratch = false;
  rc = device_radix_sort(key, A.col, A.val, key2, col2, val2, nnz, key_bits, &in_scratch, st);
  if (rc) { cleanup(); return rc; }
  if (in_scratch) { std::swap(A.col, col2); std::swap(A.val, val2); }
  rowptr.resize((size_t)rows + 1);
  t_rowptr.resize((size_t)cols + 1);
  DL(hipMemcpy(rowptr.data(), A.rowptr, sizeof(int) * (size_t)(rows + 1), hipMemcpyDeviceToHost));
  DL(hipMemcpy(t_rowptr.data(), At.rowptr, sizeof(int) * (size_t)(cols + 1), hipMemcpyDeviceToHost));
#undef DL
  cleanup();
  return 0;
}

// Both device layouts of one CSC matrix (either may be skipped: a row segment of a matrix beyond the 32-bit entry limit
// needs only one of them, create_segmented below): ingest -- on the device from 8M nonzeros, host threads below --
// then the row blocks, long-row tables and, where chosen, the sweep's tile-major copy or the column slabs.
int build_layout_pair(int dev, bool remap, bool relaxed, int64_t m, int64_t n, int64_t nnz, const int64_t *colptr,
                      const int64_t *rowval, const double *nzval, int index_base, CsrDev *A_out, CsrDev *At_out) {
  const bool verbose = getenv("PDHG_VERBOSE") != nullptr;   // phase timings of the set-up on stderr
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b2) {
    return std::chrono::duration<double>(b2 - a).count();
  };
  const auto t_start = now();
  std::vector<int> t_rowptr, rowptr;
  ivec t_col, col;
  dvec t_val, val;
  // Large matrices are laid out ON THE DEVICE (device_layout.hpp): bit-identical layouts, a fraction of
  // the host builders' time.  PDHG_DEVICE_LAYOUT=0 keeps the host builders, =1 forces the device path.
  bool device_layout = nnz >= (8 << 20) && m > 0 && n > 0;
  if (const char *ev = getenv("PDHG_DEVICE_LAYOUT")) device_layout = ev[0] != '0' && nnz > 0 && m > 0 && n > 0;
  int rc = 0;
  CsrDev dev_A, dev_At;
  if (device_layout) {
    rc = ingest_on_device(dev_A, dev_At, m, n, nnz, colptr, rowval, nzval, index_base, rowptr, t_rowptr);
    if (rc) { free_csr_dev(dev_A); free_csr_dev(dev_At); return rc; }
  } else {
    rc = csc_to_both(m, n, nnz, colptr, rowval, nzval, index_base, t_rowptr, t_col, t_val, rowptr, col, val);
    if (rc) return rc;
  }
  const auto t_conv = now();
  // the two layouts are independent: CSR(A) is built on a second host thread while this one builds CSR(A')
  // (each fans out over host_threads() workers for the per-nonzero passes; uploads are synchronous copies)
  int rc_a = 0;
  std::string err_a;
  double t_a = 0.0, t_at = 0.0;
  const int tile_a = choose_tile_cols(n, nnz, m), tile_at = choose_tile_cols(m, nnz, n);
  // Off by default: measured on the 2 x 64-core host of the GPU box at config S, the two builds side by side took
  // 0.92 s against 0.83 s one after the other (0.78 || 0.60 s against 0.47 + 0.36 s) -- the per-nonzero passes are bound
  // by host memory bandwidth, not by threads (32 threads per pass instead of 16 changed nothing either).
  const bool two = A_out && At_out && nnz >= (1 << 22) && dev_env("PDHG_PARALLEL_LAYOUTS") != nullptr;
  if (device_layout) {
    if (A_out) *A_out = dev_A; else free_csr_dev(dev_A);
    if (At_out) *At_out = dev_At; else free_csr_dev(dev_At);
    dev_A = CsrDev(); dev_At = CsrDev();
  }
  auto build_a = [&]() {
    if (!A_out) return;
    const auto t0 = now();
    if (hipSetDevice(dev) != hipSuccess) { rc_a = 999; err_a = "hipSetDevice failed on the layout thread"; return; }
    rc_a = device_layout ? build_csr_dev_resident(*A_out, (int)m, (int)n, rowptr, remap, tile_a, relaxed)
                         : build_csr_dev(*A_out, (int)m, (int)n, rowptr, col, val, remap, tile_a, relaxed);
    if (rc_a) err_a = g_last_error;
    t_a = secs(t0, now());
  };
  std::thread worker;
  if (two) worker = std::thread(build_a); else build_a();
  const auto t0 = now();
  int rc_t = 0;
  if (At_out)
    rc_t = device_layout ? build_csr_dev_resident(*At_out, (int)n, (int)m, t_rowptr, remap, tile_at, relaxed)
                         : build_csr_dev(*At_out, (int)n, (int)m, t_rowptr, t_col, t_val, remap, tile_at, relaxed);
  t_at = secs(t0, now());
  if (two) worker.join();
  if (rc_a) { g_last_error = err_a; return rc_a; }
  if (rc_t) return rc_t;
  if (verbose)
    fprintf(stderr, "pdhg_create (%s): CSC -> CSR(A), CSR(A') %.2fs; layouts%s A %.2fs %s A' %.2fs (%.2fs elapsed)\n",
            device_layout ? "device layout construction" : "host layout construction", secs(t_start, t_conv),
            device_layout ? "" : " + upload", t_a, two ? "beside" : "then", t_at, secs(t_conv, now()));
  return 0;
}

// A matrix with more entries than the layouts' 32-bit offsets index (quadratic_programming.jl:64: Int64 in the reference):
// both copies are built as SEGMENTS of whole rows (layout.hpp, CsrDev::segs) -- CSR(A) from row ranges of the matrix,
// CSR(A') from column ranges (= row ranges of A'), every range below `cap` entries -- inside ONE ordinary handle: no
// shards, no exchange, every row sum in its reference order.  Each range is ingested like a matrix of its own; the
// side of the pair that the range does not need is not built.
int build_segments(int dev, pdhg_handle *h, int64_t m, int64_t n, int64_t nnz, const int64_t *colptr, const int64_t *rowval,
                   const double *nzval, int base, int64_t cap) {
  if (!colptr || !rowval || !nzval) return fail(-1, "null input array");
  if (colptr[0] != base || colptr[n] - base != nnz) return fail(-1, "colptr does not match nnz / index_base");
  for (int64_t j = 0; j < n; ++j)
    if (colptr[j + 1] < colptr[j]) return fail(-1, "colptr not monotone");
  std::vector<int64_t> prefix;
  // (every row index is range-checked here, before anything is indexed with it)
  if (row_nnz_prefix(m, n, colptr, rowval, base, prefix) != 0) return fail(-1, "row index out of range");
  const int64_t target = std::max<int64_t>(1, (cap / 10) * 8);         // aim at 80 % of the limit
  auto cut = [&](int64_t count, auto extent, const char *what, std::vector<int64_t> &bounds) -> int {
    bounds.assign(1, 0);
    int64_t i = 0;
    while (i < count) {
      const int64_t i0 = i;
      if (extent(i0, i0 + 1) > cap)
        return fail(-2, std::string(what) + " " + std::to_string(i0) + " alone holds " + std::to_string(extent(i0, i0 + 1)) +
                            " nonzeros, more than 32-bit offsets can index (" + std::to_string(cap) + ")");
      ++i;
      while (i < count && extent(i0, i + 1) <= target) ++i;
      bounds.push_back(i);
    }
    return 0;
  };
  std::vector<int64_t> rb, cb;
  int rc;
  if ((rc = cut(m, [&](int64_t a, int64_t b) { return prefix[(size_t)b] - prefix[(size_t)a]; }, "row", rb))) return rc;
  if ((rc = cut(n, [&](int64_t a, int64_t b) { return colptr[b] - colptr[a]; }, "column", cb))) return rc;
  const bool verbose = getenv("PDHG_VERBOSE") != nullptr;
  if (verbose)
    fprintf(stderr, "[pdhg_hip] %lld nonzeros exceed the 32-bit entry limit (%lld): CSR(A) in %zu row segments, CSR(A') in %zu\n",
            (long long)nnz, (long long)cap, rb.size() - 1, cb.size() - 1);
  h->A.rows = (int)m; h->A.cols = (int)n; h->A.nnz = nnz;
  h->At.rows = (int)n; h->At.cols = (int)m; h->At.nnz = nnz;
  int slot = 0;
  for (size_t k = 0; k + 1 < rb.size(); ++k) {
    std::vector<int64_t> cp;
    uvec<int64_t> rv;
    dvec nv;
    slice_csc_rows(n, colptr, rowval, nzval, base, rb[k], rb[k + 1], cp, rv, nv);        // 0-based CSC of the row range
    CsrDev S;
    static const int64_t none_i = 0;
    static const double none_d = 0.0;
    if ((rc = build_layout_pair(dev, h->remap, h->relaxed, rb[k + 1] - rb[k], n, cp[(size_t)n], cp.data(), rv.empty() ? &none_i : rv.data(),
                                nv.empty() ? &none_d : nv.data(), 0, &S, nullptr))) { free_csr_dev(S); return rc; }
    S.row0 = (int)rb[k];
    S.slot0 = slot;
    slot += S.slots();
    h->A.max_row_nnz = std::max(h->A.max_row_nnz, S.max_row_nnz);
    h->A.segs.push_back(S);
  }
  slot = 0;
  for (size_t k = 0; k + 1 < cb.size(); ++k) {
    const int64_t c0 = cb[k], c1 = cb[k + 1], k0 = colptr[c0] - base;
    std::vector<int64_t> cp((size_t)(c1 - c0) + 1);
    for (int64_t j = c0; j <= c1; ++j) cp[(size_t)(j - c0)] = colptr[j] - colptr[c0] + base;
    CsrDev S;
    if ((rc = build_layout_pair(dev, h->remap, h->relaxed, m, c1 - c0, colptr[c1] - colptr[c0], cp.data(), rowval + k0, nzval + k0, base,
                                nullptr, &S))) { free_csr_dev(S); return rc; }
    S.row0 = (int)c0;
    S.slot0 = slot;
    slot += S.slots();
    h->At.max_row_nnz = std::max(h->At.max_row_nnz, S.max_row_nnz);
    h->At.segs.push_back(S);
  }
  return 0;
}

// One shard: device layouts + vectors for the rows it is given.  n_alloc >= n is the
// allocation length of the n-vectors that take part in collectives.
// Two of the sweep's policies are settled by TIMING the plain product on the matrix at hand, because what decides them is
// the memory regime and the builder has no cheap measure of it:
//  * the chunk variant for matrices with same-row runs of 9 ... 32 entries, 3 against 4 (spmv_kernels.hpp:
//    tiled_chunk_hybrid says why no rule on run lengths would do);
//  * whether row groups are dealt to the XCDs round robin or in contiguous eighths (tiled_per_xcd: good for bands, bad
//    when the work per row group falls with the index) -- the builder's guess (D.tw_band) against its opposite.
// One warm-up and three launches per candidate; the builder's choice stays unless another is >= 4 % faster.  Every
// candidate computes the same sums in the same order: the outcome cannot change a bit of any result, only a kernel's
// name / grid.  PDHG_TW_MODE / PDHG_TW_REMAP (dev) pin the respective choice, PDHG_TW_TUNE=0 skips the timing.
int tune_tiled_variant(pdhg_handle *h, CsrDev &D, const double *xin, double *out) {
  if (!D.tiled || D.grid <= 0 || !xin || !out) return 0;
  const char *ev = dev_env("PDHG_TW_TUNE");
  if (ev && ev[0] == '0') return 0;
  // PDHG_TUNE=0 (public): no timing at create -- the builder's static choices stand, so that kernel names, grids and
  // create time are the same from run to run and from rank to rank (pdhg_layout_describe reports what was chosen and how)
  if (const char *pv = getenv("PDHG_TUNE")) if (pv[0] == '0') return 0;
  struct Cand { int mode; bool band; float ms; };
  std::vector<Cand> cands{{D.tw_mode, D.tw_band, 0.f}};
  if (D.tw_mode == 3 && !dev_env("PDHG_TW_MODE")) cands.push_back({4, D.tw_band, 0.f});
  if (h->remap && D.grid >= 2 * NUM_XCD && !dev_env("PDHG_TW_REMAP")) cands.push_back({D.tw_mode, !D.tw_band, 0.f});
  if (cands.size() < 2) return 0;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  hipError_t ce = hipEventCreate(&e1);
  if (ce != hipSuccess) { (void)hipEventDestroy(e0); return fail((int)ce, "hipEventCreate failed"); }
  EpiArgs e{};
  e.out = out;
  int rc = 0;
  for (Cand &c : cands) {
    D.tw_mode = c.mode; D.tw_band = c.band;
    rc = launch_tiled<MODE_PLAIN>(h, D, xin, e, 0, D.grid);
    if (rc) break;
    (void)hipEventRecord(e0, h->stream);
    for (int k = 0; k < 3 && !rc; ++k) rc = launch_tiled<MODE_PLAIN>(h, D, xin, e, 0, D.grid);
    (void)hipEventRecord(e1, h->stream);
    if (!rc && hipEventSynchronize(e1) != hipSuccess) rc = fail(-3, "the sweep's tuning launches failed");
    if (rc) break;
    (void)hipEventElapsedTime(&c.ms, e0, e1);
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  D.tw_mode = cands[0].mode; D.tw_band = cands[0].band;
  if (rc) return rc;
  D.tw_tuned = (int)std::min<size_t>(cands.size(), 3);
  for (int k = 0; k < D.tw_tuned; ++k) { D.tw_tune_mode[k] = cands[(size_t)k].mode; D.tw_tune_band[k] = cands[(size_t)k].band ? 1 : 0; D.tw_tune_ms[k] = cands[(size_t)k].ms / 3.f; }
  // the chunk variant first, then the dealing (each against the builder's choice; the two are independent enough)
  size_t best = 0;
  for (size_t k = 1; k < cands.size(); ++k)
    if (cands[k].ms > 0.f && cands[k].ms < 0.96f * cands[0].ms) {
      if (cands[k].mode != cands[0].mode) D.tw_mode = cands[k].mode;
      else D.tw_band = cands[k].band;
      best = k;
    }
  (void)best;
  if (getenv("PDHG_VERBOSE")) {
    fprintf(stderr, "[pdhg_hip] sweep %d x %d timed:", D.rows, D.cols);
    for (const Cand &c : cands) fprintf(stderr, "  variant %d %s %.3f ms", c.mode, c.band ? "eighths" : "round-robin", c.ms / 3.f);
    fprintf(stderr, "  -> variant %d, %s\n", D.tw_mode, D.tw_band ? "eighths" : "round-robin");
  }
  return 0;
}

int create_shard(pdhg_handle **out, int64_t m, int64_t n, int64_t nnz,
                 const int64_t *colptr, const int64_t *rowval, const double *nzval,
                 int index_base, const double *c, const double *b, const double *lb,
                 const double *ub, int64_t num_equalities, int device_id, void *stream, int64_t n_alloc, int64_t seg_cap) {
  *out = nullptr;
  if (index_base != 0 && index_base != 1) return fail(-1, "index_base must be 0 or 1");
  if (num_equalities < 0 || num_equalities > m) return fail(-1, "num_equalities out of range");
  if (!colptr || !c || !lb || !ub || (m > 0 && !b) || (nnz > 0 && (!rowval || !nzval)))
    return fail(-1, "null input array");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev <= 0) return fail(-3, "no HIP device visible");
  int dev = device_id;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  if (dev >= ndev) return fail(-1, "device_id out of range");
  HIP_TRY(hipSetDevice(dev));
  n_alloc = std::max(n_alloc, n);

  pdhg_handle *h = new pdhg_handle();
  h->self = h;
  h->device = dev;
  h->m = m; h->n = n; h->nnz = nnz; h->num_eq = num_equalities;
  h->cn = n; h->n_alloc = n_alloc; h->m_global = m;
  const char *env = getenv("PDHG_XCD_REMAP");
  h->remap = !(env && env[0] == '0');
  env = getenv("PDHG_ROW_ORDER");         // strict: every row sum strictly left to right; relaxed (default): long rows wave-parallel
  h->relaxed = !(env && !strcmp(env, "strict"));
  env = getenv("PDHG_LAZY_ACCEPT");       // 0: pdhg_accept runs K7 itself (one more launch and n + m more words per iteration)
  h->lazy_accept = !(env && env[0] == '0');
  if (stream) { h->stream = (hipStream_t)stream; h->own_stream = false; }
  else {
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail((int)e, "hipStreamCreate failed"); }
    h->own_stream = true;
  }
#define CK(expr) do { int _rc = (expr); if (_rc) { destroy_shard(h); return _rc; } } while (0)
  if (seg_cap > 0 && nnz > seg_cap) CK(build_segments(dev, h, m, n, nnz, colptr, rowval, nzval, index_base, seg_cap));   // 64-bit extents
  else CK(build_layout_pair(dev, h->remap, h->relaxed, m, n, nnz, colptr, rowval, nzval, index_base, &h->A, &h->At));
  auto up = [&](double **dst, const double *src, int64_t len) -> int {
    int r2 = alloc_zero(dst, len);
    if (r2) return r2;
    if (len > 0) { HIP_TRY(hipMemcpy(*dst, src, sizeof(double) * (size_t)len, hipMemcpyHostToDevice)); HIP_TRY(hipStreamSynchronize(nullptr)); }
    return 0;
  };
  CK(up(&h->c, c, n)); CK(up(&h->b, b, m)); CK(up(&h->lb, lb, n)); CK(up(&h->ub, ub, n));
  CK(alloc_zero(&h->x, n_alloc)); CK(alloc_zero(&h->x_next, n_alloc)); CK(alloc_zero(&h->xbar, n_alloc));
  CK(alloc_zero(&h->y, m)); CK(alloc_zero(&h->y_next, m));
  CK(alloc_zero(&h->aty, n_alloc + 1)); CK(alloc_zero(&h->aty_next, n_alloc + 1));
  CK(alloc_zero(&h->sum_x, n)); CK(alloc_zero(&h->sum_y, m));
  CK(alloc_zero(&h->tmp_n, n_alloc)); CK(alloc_zero(&h->tmp_m, m));
  h->ew_grid_n = ew_grid(n); h->ew_grid_m = ew_grid(m); h->ew_grid_nm = ew_grid(std::max(n, m));
  h->pAt_stride = std::max(h->At.slots(), h->ew_grid_n);
  // block partials are double-double: hi parts, then lo parts
  CK(alloc_zero(&h->pA, 2 * (int64_t)std::max(h->A.slots(), 1)));
  CK(alloc_zero(&h->pAt, 6 * (int64_t)std::max(h->pAt_stride, 1)));
  CK(alloc_zero(&h->pQ, 2 * (int64_t)h->ew_grid_n));
  CK(alloc_zero(&h->scal_dev, SCAL_MAX));
  {
    hipError_t e = hipHostMalloc((void **)&h->scal_host, sizeof(double) * SCAL_MAX * DIST_MAX_WORLD, hipHostMallocDefault);
    if (e != hipSuccess) { destroy_shard(h); return fail((int)e, "hipHostMalloc failed"); }
    e = hipEventCreate(&h->ev0); if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e != hipSuccess) { destroy_shard(h); return fail((int)e, "hipEventCreate failed"); }
  }
  CK(tune_tiled_variant(h, h->A, h->xbar, h->tmp_m));
  CK(tune_tiled_variant(h, h->At, h->y, h->tmp_n));
#undef CK
  HIP_TRY(hipDeviceSynchronize());
  *out = h;
  return 0;
}

// Phase timeline of the LAST one-launch trial (PDHG_COOP_TRACE=1: the persistent kernels stamp the 100 MHz wall clock at
// every phase boundary, per workgroup).  out[0..4]: mean duration (us) over the workgroups of phase 0, barrier 1, phase 1,
// barrier 2, phase 2; out[5..9]: the slowest workgroup's; out[10]: when the last workgroup left phase 2 (us after the
// trial's first stamp); out[11]: barrier 3's global phase complete (multi-step kernel; 0: single-trial kernel);
// out[12]: decision known to the last workgroup / results published; out[13]: workgroups.
int trial_timeline(pdhg_handle *h, double out[14]) {
  if (!h->coop_trace || h->coop_grid <= 0) return 1;
  std::vector<unsigned long long> t((size_t)8 * h->coop_grid);
  if (hipMemcpy(t.data(), h->coop_trace, sizeof(unsigned long long) * t.size(), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  unsigned long long t0 = ~0ull;
  for (int w = 0; w < h->coop_grid; ++w) t0 = std::min(t0, t[(size_t)w * 8]);
  for (int k = 0; k < 5; ++k) {
    double sum = 0, mx = 0, last_end = 0;
    for (int w = 0; w < h->coop_grid; ++w) {
      const double d = 0.01 * (double)(t[(size_t)w * 8 + k + 1] - t[(size_t)w * 8 + k]);
      sum += d; mx = std::max(mx, d);
      last_end = std::max(last_end, 0.01 * (double)(t[(size_t)w * 8 + k + 1] - t0));
    }
    out[k] = sum / h->coop_grid;
    out[5 + k] = mx;
    h->timeline_last_out[k] = last_end;
  }
  out[10] = h->timeline_last_out[4];
  unsigned long long fin = 0, lead = 0;
  for (int w = 0; w < h->coop_grid; ++w) { fin = std::max(fin, t[(size_t)w * 8 + 6]); lead = std::max(lead, t[(size_t)w * 8 + 7]); }
  out[11] = lead ? 0.01 * (double)(lead - t0) : 0.0;
  out[12] = fin ? 0.01 * (double)(fin - t0) : 0.0;
  out[13] = (double)h->coop_grid;
  return 0;
}

void destroy_shard(pdhg_handle *h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->n_graph_trials > 0 && getenv("PDHG_VERBOSE"))
    fprintf(stderr, "[pdhg_hip] %ld graph trials: host us per trial: node updates %.2f, hipGraphLaunch %.2f, wait for the result %.2f\n",
            h->n_graph_trials, 1e6 * h->t_set / h->n_graph_trials, 1e6 * h->t_launch / h->n_graph_trials,
            1e6 * h->t_wait / h->n_graph_trials);
  if (h->coop_trace && (h->coop_launches > 0 || h->steps_launches > 0)) {
    // phase timeline of the LAST one-launch trial: per phase, mean and max over the workgroups of its duration (us)
    double t[14];
    if (trial_timeline(h, t) == 0) {
      const char *names[5] = {"phase 0 (x', xbar) + entry prefetch", "barrier 1", "phase 1 (A xbar, y')", "barrier 2", "phase 2 (A'y', sums)"};
      fprintf(stderr, "[pdhg_hip] one-launch trial timeline (last launch, %d workgroups, 100 MHz clock):\n", h->coop_grid);
      for (int k = 0; k < 5; ++k)
        fprintf(stderr, "    %-38s mean %6.2f us, max %6.2f us; last workgroup out at %6.2f us\n", names[k], t[k], t[5 + k], h->timeline_last_out[k]);
      if (t[11] > 0) fprintf(stderr, "    barrier 3: global phase complete at %6.2f us; decision known to the last workgroup at %6.2f us (multi-step kernel, last trial)\n",
                             t[11], t[12]);
      else fprintf(stderr, "    second-stage reduction published at %6.2f us\n", t[12]);
    }
  }
  free_csr_dev(h->A); free_csr_dev(h->At); free_csr_dev(h->Q); free_csr_dev(h->Qt);
  for (CsrDev &D : h->Achunk) free_csr_dev(D);
  if (h->chunk_carry) (void)hipFree(h->chunk_carry);
  if (h->xchunk) (void)hipFree(h->xchunk);
  for (hipEvent_t ev : h->ev_ag) if (ev) (void)hipEventDestroy(ev);
  if (h->ev_xbar) (void)hipEventDestroy(h->ev_xbar);
  double *bufs[] = {h->c, h->b, h->lb, h->ub, h->x, h->x_next, h->xbar, h->y, h->y_next,
                    h->aty, h->aty_next, h->sum_x, h->sum_y, h->qx, h->tmp_n, h->tmp_n2,
                    h->tmp_m, h->pA, h->pAt, h->pQ, h->scal_dev, h->scal_all, h->dn_buf, h->dm_buf,
                    h->E, h->Dv, h->c_o, h->b_o, h->lb_o,
                    h->ub_o, h->x_r, h->y_r, h->px_avg, h->py_avg, h->ev_ax, h->ev_aty, h->tr_g,
                    h->tr_dir, h->tr_thr, h->ev_partials, h->ev_cax[0], h->ev_cax[1], h->ev_cax[2],
                    h->ev_caty[0], h->ev_caty[1], h->ev_caty[2], h->ev_cqx[0], h->ev_cqx[1], h->ev_cqx[2],
                    h->ev_qx, h->ev_xg};
  for (double *p : bufs) if (p) (void)hipFree(p);
  graph_destroy(h->tgraph[0]); graph_destroy(h->tgraph[1]);
  if (h->comm_stream) { (void)hipStreamSynchronize(h->comm_stream); (void)hipStreamDestroy(h->comm_stream); }
  for (hipEvent_t ev : h->ev_part) if (ev) (void)hipEventDestroy(ev);
  if (h->ev_comm) (void)hipEventDestroy(h->ev_comm);
  if (h->seq_dev) (void)hipFree(h->seq_dev);
  if (h->gsync) (void)hipFree(h->gsync);
  if (h->tr_sync) (void)hipFree(h->tr_sync);
  if (h->lsync) (void)hipFree(h->lsync);
  if (h->tr_partials) (void)hipFree(h->tr_partials);
  if (h->trb_scratch) (void)hipFree(h->trb_scratch);
  if (h->trb_partials) (void)hipFree(h->trb_partials);
  if (h->coop_trace) (void)hipFree(h->coop_trace);
  if (h->res_host) (void)hipHostFree((void *)h->res_host);
  if (h->scal_host) (void)hipHostFree(h->scal_host);
  if (h->ev_host) (void)hipHostFree(h->ev_host);
  if (h->steps_ctl) (void)hipFree(h->steps_ctl);
  if (h->steps_pow_dev) (void)hipFree(h->steps_pow_dev);
  if (h->steps_pow_host) (void)hipHostFree(h->steps_pow_host);
  if (h->steps_res) (void)hipHostFree(h->steps_res);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

void destroy_group(DistGroup *g) {
  if (!g) return;
  if (g->n_trials > 0 && getenv("PDHG_VERBOSE"))
    fprintf(stderr, "[pdhg_hip] %lld group trials over %zu local shards (%s): host us per trial: issuing %.1f, waiting for the scalars %.1f\n",
            (long long)g->n_trials, g->sh.size(), g->pool ? "one issuing thread per shard" : "issued by the calling thread",
            1e6 * g->t_issue / g->n_trials, 1e6 * g->t_wait / g->n_trials);
  delete g->pool;
  g->pool = nullptr;
  for (size_t i = 0; i < g->sh.size(); ++i) {
    (void)hipSetDevice(g->sh[i]->device);
    (void)hipStreamSynchronize(g->sh[i]->stream);
  }
  for (ncclComm_t c : g->comm) if (c) (void)rccl_loader().api.CommDestroy(c);   // a communicator exists only if RCCL was bound
  for (int f = 0; f < 2; ++f)
    for (size_t i = 0; i < g->ev[f].size(); ++i) {
      (void)hipSetDevice(g->sh[i]->device);
      if (g->ev[f][i]) (void)hipEventDestroy(g->ev[f][i]);
    }
  group_coop_release(*g);
  if (g->gsync) { (void)hipSetDevice(g->sh.empty() ? 0 : g->sh[0]->device); (void)hipFree(g->gsync); }
  for (pdhg_handle *s : g->sh) destroy_shard(s);
  delete g;
}

int create_rank_shard_local(DistGroup *g, int rank, int64_t n, const int64_t *colptr, const int64_t *rowval,
                            const double *nzval, int base, const double *c, const double *b_local, const double *lb,
                            const double *ub, int device_id, void *stream, pdhg_handle **out);
int build_column_chunks(DistGroup *g, pdhg_handle *s, int64_t n, const int64_t *colptr, const int64_t *rowval,
                        const double *nzval, int base);

// Build rank `rank`'s shard of the GLOBAL problem: rows row_lo[rank]..row_lo[rank+1), all columns.
int create_rank_shard(DistGroup *g, int rank, int64_t n, const int64_t *colptr, const int64_t *rowval,
                      const double *nzval, int base, const double *c, const double *b, const double *lb,
                      const double *ub, int device_id, void *stream, pdhg_handle **out) {
  const int64_t lo = g->row_lo[(size_t)rank], hi = g->row_lo[(size_t)rank + 1];
  std::vector<int64_t> cp;
  uvec<int64_t> rv;
  dvec nv;
  slice_csc_rows(n, colptr, rowval, nzval, base, lo, hi, cp, rv, nv);
  return create_rank_shard_local(g, rank, n, cp.data(), rv.data(), nv.data(), 0, c, b ? b + lo : nullptr, lb, ub,
                                 device_id, stream, out);
}

// The same from the rank's OWN rows: (colptr, rowval, nzval) is the CSC of rows lo..hi of the
// global matrix with row indices rebased to 0, b_local its hi - lo right-hand sides.
int create_rank_shard_local(DistGroup *g, int rank, int64_t n, const int64_t *colptr, const int64_t *rowval,
                            const double *nzval, int base, const double *c, const double *b_local, const double *lb,
                            const double *ub, int device_id, void *stream, pdhg_handle **out) {
  const int64_t lo = g->row_lo[(size_t)rank], hi = g->row_lo[(size_t)rank + 1];
  const int64_t ne = std::min<int64_t>(std::max<int64_t>(g->num_eq_global - lo, 0), hi - lo);
  pdhg_handle *s = nullptr;
  int rc = create_shard(&s, hi - lo, n, colptr[n] - base, colptr, rowval, nzval, base, c, b_local,
                        lb, ub, ne, device_id, stream, g->world * g->S, 0);
  if (rc) return rc;
  if (hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&s->ev_comm, hipEventDisableTiming) != hipSuccess) {
    destroy_shard(s);
    return fail(999, "comm stream / event creation failed");
  }
  s->ev_part.assign((size_t)g->world, nullptr);
  for (int k = 0; k < g->world; ++k)
    if (hipEventCreateWithFlags(&s->ev_part[(size_t)k], hipEventDisableTiming) != hipSuccess) {
      destroy_shard(s);
      return fail(999, "event creation failed");
    }
  s->grp = g;
  s->rank = rank;
  s->world = g->world;
  s->row_lo = lo;
  s->m_global = g->m_global;
  s->clo = std::min<int64_t>(n, (int64_t)rank * g->S);
  s->cn = std::min<int64_t>(n, (int64_t)(rank + 1) * g->S) - s->clo;
  if ((rc = alloc_zero(&s->dn_buf, s->n_alloc))) { destroy_shard(s); return rc; }
  if ((rc = alloc_zero(&s->dm_buf, g->m_global))) { destroy_shard(s); return rc; }
  if (g->ag_chunks > 1 && (rc = build_column_chunks(g, s, n, colptr, rowval, nzval, base))) { destroy_shard(s); return rc; }
  if ((rc = alloc_zero(&s->scal_all, (int64_t)SCAL_MAX * g->world))) { destroy_shard(s); return rc; }
  *out = s;
  return 0;
}

// The column-chunk layouts of one shard (DistGroup::ag_chunks, dist.hpp): chunk c holds the entries of A_p whose column lies
// in sub-range c of its owner's slice, as a complete layout of its own -- rows x (world * ag_sub) with column indices INTO
// THE CHUNK (rank after rank, what one all-gather of the chunk delivers: pdhg_handle::xchunk), no column slabs (the
// passes carry the row sums themselves) -- built by the ordinary builder from the filtered, re-indexed CSC arrays.
int build_column_chunks(DistGroup *g, pdhg_handle *s, int64_t n, const int64_t *colptr, const int64_t *rowval,
                        const double *nzval, int base) {
  const int C = g->ag_chunks;
  const int64_t m = s->m, sub = g->ag_sub, W = (int64_t)g->world * sub;       // W: columns of a chunk (chunk layout, dist.hpp)
  int rc = 0;
  s->Achunk.resize((size_t)C);
  std::vector<int64_t> cp((size_t)W + 1);
  uvec<int64_t> rv;
  dvec nv;
  // chunk column jc = q * sub + off  <->  natural column j = q * S + c * sub + off (ascending jc = ascending j inside a chunk:
  // the passes add a row's products of one chunk in ascending column order)
  auto natural = [&](int c, int64_t jc) { const int64_t q = jc / sub, off = jc - q * sub; const int64_t i = (int64_t)c * sub + off; return i < g->S ? q * g->S + i : (int64_t)-1; };
  for (int c = 0; c < C && !rc; ++c) {
    int64_t cnt = 0;
    for (int64_t jc = 0; jc < W; ++jc) {
      cp[(size_t)jc] = cnt;
      const int64_t j = natural(c, jc);
      if (j >= 0 && j < n) cnt += colptr[j + 1] - colptr[j];
    }
    cp[(size_t)W] = cnt;
    rv.resize((size_t)std::max<int64_t>(cnt, 1));
    nv.resize((size_t)std::max<int64_t>(cnt, 1));
    parallel_ranges((int)std::min<int64_t>(W, INT32_MAX), 1 << 14, [&](int jb, int je) {
      for (int64_t jc = jb; jc < je; ++jc) {
        const int64_t j = natural(c, jc);
        if (j < 0 || j >= n) continue;
        const int64_t k0 = colptr[j] - base, k1 = colptr[j + 1] - base, d0 = cp[(size_t)jc];
        for (int64_t k = k0; k < k1; ++k) { rv[(size_t)(d0 + k - k0)] = rowval[k] - base; nv[(size_t)(d0 + k - k0)] = nzval[k]; }
      }
    });
    g_no_slabs = true;
    rc = build_layout_pair(s->device, s->remap, s->relaxed, m, W, cnt, cp.data(), rv.data(), nv.data(), 0, &s->Achunk[(size_t)c], nullptr);
    g_no_slabs = false;
  }
  if (!rc) rc = alloc_zero(&s->xchunk, (int64_t)C * W);
  if (rc) return rc;
  if ((rc = alloc_zero(&s->chunk_carry, std::max<int64_t>(m, 1)))) return rc;
  // the last pass's block partials: pA must hold its slots too
  if (dual_chunk_slots(s) > s->A.slots()) {
    (void)hipFree(s->pA);
    s->pA = nullptr;
    if ((rc = alloc_zero(&s->pA, 2 * (int64_t)dual_chunk_slots(s)))) return rc;
  }
  s->ev_ag.assign((size_t)C, nullptr);
  for (int c = 0; c < C; ++c)
    if (hipEventCreateWithFlags(&s->ev_ag[(size_t)c], hipEventDisableTiming) != hipSuccess) return fail(999, "event creation failed");
  if (hipEventCreateWithFlags(&s->ev_xbar, hipEventDisableTiming) != hipSuccess) return fail(999, "event creation failed");
  return 0;
}

// One reduce-scatter after the product, or per-slice reductions overlapped with it
// (DistGroup::overlap).  Decided from (n, world, back end, environment) only.  Default: on
// for the peer back end when the exchanged vector is large (a slice is one small kernel of
// its owner); OFF for RCCL, where P reductions to P roots cost P collective latencies and
// may not reach the bandwidth of one reduce-scatter over all links -- the product they
// could hide behind is 0.1 ms at P = 8.  PDHG_DIST_OVERLAP=0/1 forces.
// The all-gather side (DistGroup::ag_chunks): PDHG_DIST_AG_OVERLAP=1 cuts xbar's all-gather and A_p xbar into
// PDHG_DIST_AG_CHUNKS (4) column chunks, pass c running beside the transfer of chunk c + 1; =2 runs the same passes behind
// one all-gather (the bitwise reference of =1; what the peer back end does under either value).  Default OFF: RCCL's
// kernels beside a sweep that fills every compute unit have never run on hardware.
void choose_exchange_pattern(DistGroup *g) {
  const char *ov = getenv("PDHG_DIST_OVERLAP");
  g->overlap = ov ? (ov[0] != '0') : (g->backend == COMM_P2P && g->world > 1 && g->n * 8 > (4LL << 20));
  const char *ag = getenv("PDHG_DIST_AG_OVERLAP");
  g->ag_mode = (ag && (ag[0] == '1' || ag[0] == '2') && !ag[1]) ? ag[0] - '0' : 0;
  g->ag_chunks = 0;
  if (g->ag_mode && g->world > 1) {
    const char *cv = dev_env("PDHG_DIST_AG_CHUNKS");
    int C = cv ? std::max(2, std::min(16, atoi(cv))) : 4;
    int64_t sub = ((g->S + C - 1) / C + 15) / 16 * 16;         // whole 128-byte lines
    C = (int)((g->S + sub - 1) / sub);                        // (a short slice may give fewer chunks)
    if (C > 1) { g->ag_chunks = C; g->ag_sub = sub; }
  }
  if (g->ag_chunks <= 1) { g->ag_chunks = 0; g->ag_mode = 0; }
}

// row_bounds != nullptr: the caller's partition ([world + 1], ascending, 0 .. m) instead of the
// library's nnz-balanced one (colptr / rowval may then be null)
int init_group_geometry(DistGroup *g, int64_t m, int64_t n, const int64_t *colptr, const int64_t *rowval, int base,
                        int64_t num_equalities, int world, const int64_t *row_bounds = nullptr) {
  if (world < 1 || world > DIST_MAX_WORLD) return fail(-1, "world size out of range (1..64)");
  if (!colptr && !row_bounds) return fail(-1, "null input array");
  g->world = world;
  g->n = n;
  g->m_global = m;
  g->num_eq_global = num_equalities;
  const int64_t per = (n + world - 1) / world;
  g->S = std::max<int64_t>(16, (per + 15) / 16 * 16);      // slice stride: whole 128-byte lines
  if (row_bounds) {
    if (row_bounds[0] != 0 || row_bounds[world] != m) return fail(-1, "row_bounds must run from 0 to m");
    for (int p = 0; p < world; ++p)
      if (row_bounds[p + 1] < row_bounds[p]) return fail(-1, "row_bounds not ascending");
    g->row_lo.assign(row_bounds, row_bounds + world + 1);
  } else {
    partition_rows_by_nnz(m, n, colptr, rowval, base, world, g->row_lo);
  }
  const char *fr = dev_env("PDHG_DIST_FORCE_REMOTE");
  g->force_remote = fr && fr[0] == '1';
  return 0;
}

