// layout.hpp -- part of the single translation unit pdhg_hip.hip (included there, in order).
// Host side of the device layouts: CsrDev, uploads, the stream/tiled builders.
#pragma once

namespace {

// ---------------------------------------------------------------- host side

// One column slab of a stream-layout matrix: its own CSR (absolute column indices, rows
// that are "long" in the full matrix left empty) and row blocks.
struct SlabDev {
  int *rowptr = nullptr, *col = nullptr;
  double *val = nullptr;
  int2 *blks = nullptr;
  int nblk = 0, per_xcd = 0, grid = 0;
  int64_t nnz = 0;
  SjDev sj;                        // the slab's entries in the sliced jagged layout (sj_kernels.hpp), when built
  int4 *ext = nullptr;             // (r0, r1, k0, k1) per row block + the persistent kernel's grid (spmv_stream_pipe_kernel), when built
  int pipe_grid = 0;
  CsrView view(int rows) const { return CsrView{rows, rowptr, col, val}; }
};

struct CsrDev {
  int rows = 0, cols = 0;
  int64_t nnz = 0;
  int *rowptr = nullptr, *col = nullptr;
  double *val = nullptr;
  int2 *blks = nullptr;
  int nblk = 0, per_xcd = 0, grid = 0;
  int nlong = 0, nchunks = 0, long_grid = 0;      // long_grid: workgroups of the long-row final kernel (4 rows each)
  int *long_row = nullptr, *long_chunk_ptr = nullptr, *chunk_row = nullptr, *chunk_off = nullptr;
  int *chunk_lidx = nullptr;                 // [nchunks] index of the chunk's row in long_row
  unsigned long long *long_ticket = nullptr; // [nlong] chunks of the row completed, over all one-launch trials
  unsigned long long coop_uses = 0;          // one-launch trials that have run this layout's product (the tickets' base)
  double *chunk_partial = nullptr;
  int64_t max_row_nnz = 0;
  int long_thr = BLOCK_NNZ;          // rows beyond this many entries: long-row kernels (common.hpp)
  // tiled-sweep layout (optional)
  bool tiled = false;
  int tile_shift = 0, tile_cols = 0, nwaves = 0, ntiles = 0, tw_rows = 0;   // tile widths <= 1 << tile_shift
  size_t tw_lds_floor = 0;         // dynamic LDS is padded to this: keeps a third workgroup off the CU
  bool var_tiles = false;          // equal-nonzero tiles of different widths (skewed columns); tile_cols is then the nominal width
  int2 *wave_rows = nullptr;
  int *wave_ent = nullptr;        // per-wave entry offsets, one per step of its workgroup (+1)
  int *wave_step_off = nullptr;   // [nwaves] start of a wave's offsets inside wave_ent
  int *step_tile = nullptr;       // first column of the tile of every step, workgroup after workgroup
  int *wg_step_off = nullptr;     // [grid+1] start of a workgroup's steps inside step_tile
  int64_t total_steps = 0;
  int64_t tw_entries = 0, step_ptr_len = 0;   // lengths of pk / tv and of wave_ent (checksums, tests)
  double tw_touched = 1.0;        // share of the sweep's (workgroup, tile) cells that hold entries, as build_tiled last measured it (also when it declined)
  bool tw_band = false;           // a workgroup touches < 90 % of the tiles (banded / block-local rows): row groups dealt to the XCDs in contiguous eighths
  int tw_tuned = 0;               // tune_tiled_variant timed candidates on this matrix at create: how many; their times (ms per product) below
  int tw_tune_mode[3] = {0, 0, 0}, tw_tune_band[3] = {0, 0, 0};
  float tw_tune_ms[3] = {0.f, 0.f, 0.f};
  int tw_mode = 0;                // chunk accumulation: 0 lane shuffles, 1 LDS scratch (long runs, strict order), 2 relaxed order, 3 lane to lane (runs of 9 ... 32), 4 = 3 / 0 per chunk (chosen over 3 by timing: tune_tiled_variant)
  unsigned *pk = nullptr;
  double *tv = nullptr;
  std::vector<int> wg_first_row;  // host copy: first row of every tiled workgroup (+ rows), for partial launches
  // column-slab passes (optional, stream layout only): when non-empty the product runs as one
  // stream-kernel launch per slab and `grid` is the LAST slab's grid (its blocks write the partials)
  std::vector<SlabDev> slabs;
  double *slab_partial = nullptr;   // [rows] row sums between the passes
  // sliced jagged copy of the stream layout (sj_kernels.hpp): the product kernel of stream-class matrices with more row
  // blocks than the persistent trial kernels take (the CSR arrays and row blocks above stay: evaluation-time callers,
  // the one-launch paths and the long-row kernels use them)
  SjDev sj;
  // the CSR row blocks as a persistent software-pipelined launch (spmv_stream_pipe_kernel): extent words + grid, when built
  int4 *ext = nullptr;
  int pipe_grid = 0;
  // ---- 64-bit extents (quadratic_programming.jl:64: the reference's indices are Int64).  The kernels index entries with
  // 32 bits; a matrix with more entries than that is held as SEGMENTS of whole consecutive rows, each a complete CsrDev
  // of its own (own arrays and tables, offsets local to the segment: the 64-bit part of an entry's address is the
  // segment's base pointer).  This head then owns no arrays: rows / cols / nnz are the whole matrix's, `segs` the
  // parts in row order.  Rows are never cut, so every row sum keeps its order and its bits, and the two products need no
  // exchange (unlike the row SHARDS of dist.hpp, whose A_p' products are partial sums).
  std::vector<CsrDev> segs;
  int row0 = 0;                    // a segment's first row in the whole matrix
  int slot0 = 0;                   // ... and its first block-partial slot
  int slots() const {              // block partials: one per row block, one per long row
    if (segs.empty()) return grid + nlong;
    int total = 0;
    for (const CsrDev &S : segs) total += S.slots();
    return total;
  }
  CsrView view() const { return CsrView{rows, rowptr, col, val}; }
};

// Big host arrays (one element per nonzero) must not be value-initialised: the fill is
// serial and, at a billion nonzeros, costs more than the threaded passes that then
// overwrite every element (first touch happens in those passes, spread over the threads).
template <class T>
struct default_init_allocator : std::allocator<T> {
  template <class U> struct rebind { using other = default_init_allocator<U>; };
  template <class U, class... Args>
  void construct(U *p, Args &&...args) {
    if constexpr (sizeof...(args) == 0) ::new ((void *)p) U;
    else ::new ((void *)p) U(std::forward<Args>(args)...);
  }
};
template <class T> using uvec = std::vector<T, default_init_allocator<T>>;
typedef uvec<int> ivec;
typedef uvec<double> dvec;

template <typename T, typename Alloc>
int upload(T **dst, const std::vector<T, Alloc> &src) {
  const size_t bytes = sizeof(T) * std::max<size_t>(src.size(), 1);
  HIP_TRY(hipMalloc((void **)dst, bytes));
  if (!src.empty()) {
    HIP_TRY(hipMemcpy(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipStreamSynchronize(nullptr));
  }
  return 0;
}

int alloc_zero(double **dst, int64_t len) {
  const size_t bytes = sizeof(double) * (size_t)std::max<int64_t>(len, 1);
  HIP_TRY(hipMalloc((void **)dst, bytes));
  // the fill runs on the null stream, which does NOT order against the handles'
  // non-blocking streams: wait for it, or a kernel launched next on such a stream
  // can have its output zeroed by a late fill
  HIP_TRY(hipMemsetAsync(*dst, 0, bytes, nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));
  return 0;
}

// Run f(begin, end) over [0, n) cut into contiguous ranges, one host thread per
// range (PDHG_HOST_THREADS, default min(16, hardware threads)); fewer threads when
// a range would hold less than `grain` items, inline when one is enough.
template <typename F>
void parallel_ranges(int n, int grain, F f) {
  int threads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char *ev = getenv("PDHG_HOST_THREADS")) threads = std::max(1, atoi(ev));
  threads = std::min(threads, std::max(1, n / std::max(1, grain)));
  if (threads <= 1) { f(0, n); return; }
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    const int b = (int)((int64_t)n * t / threads), e = (int)((int64_t)n * (t + 1) / threads);
    pool.emplace_back([=, &f] { f(b, e); });
  }
  for (std::thread &th : pool) th.join();
}

// Widest tile the L2 holds against the entry stream (measured, profiles/r02_tile_rule.txt):
// 76K columns when several residency rounds of the sweep are in flight, 144K for one round.
inline int tile_width_cap(int64_t rows) {
  const int64_t slots = 256LL * 2 * TW_WPB;
  const int64_t rounds = std::max<int64_t>(1, (rows + slots * TW_MAX_ROWS - 1) / (slots * TW_MAX_ROWS));
  return (rounds == 1 ? 144 : 76) * 1024;
}

// Host-side construction of the tiled-sweep layout: wave row blocks (runs of
// <= TW_ROWS consecutive non-long rows) and their entries counting-sorted by
// column tile (stable, so (row, col) order is kept inside a tile).
// Device mode (col_p == nullptr; device_layout.hpp): D.rowptr / D.col / D.val are already in HBM.  The
// three places that touch the entries -- the 16-column histogram of the skew test, the
// per-(wave, tile) counts and the tile-major fill -- run as kernels there (tw_count_kernel,
// cnt16_kernel, tw_fill_kernel) and everything else is the code below, unchanged: same plan, same
// tables, bit-identical layout.  Returns 1 (nothing built, D untouched) when the device mode does
// not cover the case -- equal-nonzero tiles of different widths, more than 1024 tiles, a count
// matrix beyond 256M cells: the caller then fetches the entries and calls the host mode.
int build_tiled(CsrDev &D, int rows, const std::vector<int> &rowptr, const ivec *col_p,
                const dvec *val_p, int tile_cols, bool relaxed) {
  const bool on_device = col_p == nullptr;
  static const ivec no_col;
  static const dvec no_val;
  const ivec &col = on_device ? no_col : *col_p;
  const dvec &val = on_device ? no_val : *val_p;
  // Tile boundaries.  Uniform width tile_cols by default.  When the columns are skewed
  // (hub columns: the fullest uniform tile holds more than 1.5x the average), the
  // boundaries are moved so that every tile holds about the same number of entries
  // (widths are multiples of 16 columns, at most the L2 cap): a hub region gets narrow
  // tiles instead of cells cut into many barrier-separated sub-steps, and the entries a
  // row has inside one tile -- summed by one lane, in order -- stay few.
  std::vector<int> tstart;                  // [ntiles + 1] first column of every tile
  std::vector<int> map16;                   // variable widths: tile of a 16-column group
  {
    const int nt0 = std::max<int>(1, (int)((((int64_t)D.cols) + tile_cols - 1) / tile_cols));
    // entries per 16-column group, on host threads (integer counts: the result does not depend on the thread count)
    const int groups = (D.cols + 15) / 16;
    std::vector<int> cnt16((size_t)groups, 0);
    if (on_device) {
      int *d16 = nullptr;
      HIP_TRY(hipMalloc((void **)&d16, sizeof(int) * (size_t)std::max(groups, 1)));
      HIP_TRY(hipMemset(d16, 0, sizeof(int) * (size_t)std::max(groups, 1)));
      hipLaunchKernelGGL(cnt16_kernel, dim3(2048), dim3(TPB), 0, nullptr, rows, (const int *)D.rowptr, (const int *)D.col, d16, D.long_thr);
      hipError_t e = hipMemcpy(cnt16.data(), d16, sizeof(int) * (size_t)groups, hipMemcpyDeviceToHost);
      (void)hipFree(d16);
      HIP_TRY(e);
    } else {
      std::mutex merge;
      parallel_ranges(rows, 1 << 16, [&](int rb, int re) {
        std::vector<int> mine((size_t)groups, 0);
        for (int r = rb; r < re; ++r) {
          if (rowptr[r + 1] - rowptr[r] > D.long_thr) continue;    // long rows are not in the sweep
          for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) mine[(size_t)(col[k] >> 4)] += 1;
        }
        std::lock_guard<std::mutex> lock(merge);
        for (int g = 0; g < groups; ++g) cnt16[(size_t)g] += mine[(size_t)g];
      });
    }
    std::vector<int64_t> hist((size_t)nt0, 0);
    int64_t total = 0;
    for (int g = 0; g < groups; ++g) {
      // a group straddling two uniform tiles is counted with its first column's tile: a skew test, not a layout
      hist[(size_t)(((int64_t)g * 16) / tile_cols)] += cnt16[(size_t)g];
      total += cnt16[(size_t)g];
    }
    int64_t fullest = 0;
    for (int64_t v : hist) fullest = std::max(fullest, v);
    const char *vt = dev_env("PDHG_VAR_TILES");                   // 0 / 1 force
    bool variable = nt0 > 1 && (double)fullest > 1.5 * (double)total / (double)nt0;
    if (vt) variable = vt[0] != '0' && nt0 > 1;
    if (on_device && nt0 > 1024) return 1;                    // host mode handles these
    if (!variable) {
      for (int t = 0; t <= nt0; ++t) tstart.push_back((int)std::min<int64_t>(D.cols, (int64_t)t * tile_cols));
    } else {
      const int64_t target = std::max<int64_t>(1, total / nt0);
      const int wmax = std::max(16, std::min(tile_width_cap(rows), 2 * tile_cols) / 16 * 16);
      map16.assign((size_t)groups, 0);
      tstart.push_back(0);
      int64_t have = 0;
      int g0 = 0;
      for (int g = 0; g < groups; ++g) {
        map16[(size_t)g] = (int)tstart.size() - 1;
        have += cnt16[(size_t)g];
        const bool last = g + 1 == groups;
        if (last || have >= target || (g + 1 - g0) * 16 >= wmax) {
          tstart.push_back(last ? D.cols : (g + 1) * 16);
          have = 0;
          g0 = g + 1;
        }
      }
      if ((int64_t)tstart.size() - 1 > 65536) return 0;          // absurdly many tiles: leave it to the stream layout
    }
  }
  const int ntiles = (int)tstart.size() - 1;
  if (on_device && ntiles > 1024) return 1;                   // the kernels keep one counter per tile and wave in LDS
  int widest = 1;
  for (int t = 0; t < ntiles; ++t) widest = std::max(widest, tstart[(size_t)t + 1] - tstart[(size_t)t]);
  // the column field of an entry is wide enough for the widest tile (any width, not only powers of two)
  int tile_shift = 1;
  while ((1LL << tile_shift) < widest) ++tile_shift;
  const bool uniform = map16.empty();
  const bool pow2 = uniform && (1LL << tile_shift) == tile_cols;
  const int *map16p = map16.data();
  auto tile_of = [=](int c) { return !uniform ? map16p[c >> 4] : (pow2 ? (c >> tile_shift) : (c / tile_cols)); };
  // Geometry.  A CU holds 2 workgroups of 8 waves; the grid runs in rounds of
  // 256 CUs x 16 waves.  Rows per wave is chosen so that the rounds are full
  // (no tail round), within the LDS budget (160 KiB / 16 waves).
  // dev knobs (tools/tune_tiled.py): workgroups per CU the geometry is planned for, rows per wave cap
  const char *wpc_env = dev_env("PDHG_TW_WGS_PER_CU"), *mr_env = dev_env("PDHG_TW_MAX_ROWS");
  const int wgs_per_cu = wpc_env ? std::max(1, atoi(wpc_env)) : 2;
  const int64_t slots = 256LL * wgs_per_cu * TW_WPB;      // resident waves per round
  // row_local must stay below all-ones in its bit field: {row_local << shift | col_local}
  // == TW_PAD (0xFFFFFFFF) would be taken for padding and dropped.
  const int max_rows = std::min<int>(mr_env ? std::max(64, atoi(mr_env)) : TW_MAX_ROWS, (1 << (32 - tile_shift)) - 1);
  int TW_ROWS;
  {
    int64_t rounds = std::max<int64_t>(1, ((int64_t)rows + slots * max_rows - 1) / (slots * max_rows));
    int64_t rpw = ((int64_t)rows + slots * rounds - 1) / (slots * rounds);
    TW_ROWS = (int)std::max<int64_t>(64, std::min<int64_t>(max_rows, rpw));
  }
  const bool rows_forced = dev_env("PDHG_TW_ROWS") != nullptr;
  if (rows_forced) TW_ROWS = std::max(1, std::min(atoi(dev_env("PDHG_TW_ROWS")), max_rows));
  const int WIN = TW_U * WAVE;  // entries a wave holds in registers per step
  // pass 1: wave row blocks.  A wave owns <= TW_ROWS rows AND <= nnz_cap
  // nonzeros: hub regions (PageRank's oldest nodes) would otherwise give one
  // wave 100x the average work and the whole launch would wait for its workgroup.
  // The entry cap (and skipped long rows) can leave a few waves more than fill the
  // planned residency rounds; one workgroup beyond them runs a whole extra sweep by
  // itself (measured: 8 195 waves instead of 8 192 cost +45 %).  Rows per wave are
  // then raised, within the LDS budget (and, on a near miss, the entry cap a little),
  // until the waves fit the rounds again.
  const bool cap_forced = dev_env("PDHG_TW_NNZ_CAP") != nullptr;
  double cap_factor = cap_forced ? std::max(1.0, atof(dev_env("PDHG_TW_NNZ_CAP"))) : 2.0;   // dev knob
  bool balanced = false;
  const int ntiles_planned = (int)tstart.size() - 1;
  std::vector<int2> wave_rows;
  for (int attempt = 0;; ++attempt) {
    const int64_t est_waves = std::max<int64_t>(1, ((int64_t)rows + TW_ROWS - 1) / TW_ROWS);
    const int64_t nnz_cap = std::max<int64_t>(dev_env("PDHG_TW_NNZ_CAP") ? 256 : 4096,
                                              (int64_t)(cap_factor * (double)(D.nnz / est_waves)));   // 2x the average wave
    wave_rows.clear();
    int r = 0;
    while (r < rows) {
      if (rowptr[r + 1] - rowptr[r] > D.long_thr) { ++r; continue; }  // long row: separate path
      const int r0 = r;
      while (r < rows && (r - r0) < TW_ROWS && rowptr[r + 1] - rowptr[r] <= D.long_thr &&
             (r == r0 || (int64_t)rowptr[r + 1] - rowptr[r0] <= nnz_cap)) ++r;
      wave_rows.push_back(make_int2(r0, r));
    }
    const int64_t planned = slots * std::max<int64_t>(1, ((int64_t)est_waves + slots - 1) / slots);
    if (rows_forced || (int64_t)wave_rows.size() <= planned || attempt >= 40 || balanced) break;
    if (TW_ROWS < max_rows) { TW_ROWS = std::min(max_rows, TW_ROWS + std::max(1, TW_ROWS / 48)); continue; }
    if (cap_forced) break;
    // A near miss: let the waves at the entry cap grow a little -- as long as their (wave, tile) cells still fit the
    // register window (with a quarter's margin for the spread): a cell beyond it costs the whole workgroup a second step
    // on that tile.  When they would not (rows of very different lengths: many waves sit at the cap), the extra
    // residency round is taken and the waves are BALANCED instead -- cap 1.3 x the average.  Column-skewed 10M, A'y'
    // (tools/archive/r5_colskew_sweep.sh): cap grown to 2.9 to fit two rounds 1.64 ms (257 steps for the heavy workgroups, at 4.9 us
    // each: long same-row runs); cap 1.0 / 1.3 / 1.6, three rounds of 129 steps, 1.13 / 1.11 / 1.11 ms.
    const double avg_cell = (double)D.nnz / (double)est_waves / (double)std::max(1, ntiles_planned);
    const bool near_miss = (double)wave_rows.size() <= 1.03 * (double)planned && cap_factor < 4.0;
    if (near_miss && 1.2 * cap_factor * avg_cell * 1.25 <= (double)WIN) cap_factor *= 1.2;
    else if (near_miss) { cap_factor = 1.3; balanced = true; }
    else break;
  }
  D.tw_rows = TW_ROWS;
  const int nwaves = (int)wave_rows.size();
  const int grid = (nwaves + TW_WPB - 1) / TW_WPB;
  // The per-workgroup work below (count the cells, cut heavy ones into steps,
  // scatter the entries tile-major) is independent from workgroup to workgroup
  // once the output offsets are known, so it runs on host threads in two passes
  // (A: sizes, serial prefix sums, B: fill).  The result does not depend on the
  // number of threads.
  std::vector<int> wg_nsteps((size_t)std::max(grid, 1), 0), wg_touched((size_t)std::max(grid, 1), 0);
  // device mode: the per-(wave, tile) counts and the longest same-row run come from one kernel pass
  std::vector<int> cell_cnt, dev_max_run;
  int2 *d_wave_rows = nullptr;
  int *d_cell_cnt = nullptr;
  struct DevTmp {          // freed on every exit unless handed over
    void **p;
    ~DevTmp() { if (*p) (void)hipFree(*p); }
  } tmp_wave_rows{(void **)&d_wave_rows}, tmp_cell_cnt{(void **)&d_cell_cnt};
  int *d_map16 = nullptr, *d_tstart = nullptr;                // tiles of different widths: the kernels' lookup tables
  DevTmp tmp_map16{(void **)&d_map16}, tmp_tstart{(void **)&d_tstart};
  if (on_device && !uniform) {
    int rc2;
    if ((rc2 = upload(&d_map16, map16))) return rc2;
    if ((rc2 = upload(&d_tstart, tstart))) return rc2;
  }
  if (on_device) {
    if ((int64_t)nwaves * ntiles > (256LL << 20)) return 1;     // the count matrix travels to the host: <= 1 GiB
    cell_cnt.resize((size_t)std::max(nwaves, 1) * ntiles);
    dev_max_run.assign((size_t)std::max(grid, 1), 0);
    int rc2;
    if ((rc2 = upload(&d_wave_rows, wave_rows))) return rc2;
    int *d_mr = nullptr;
    HIP_TRY(hipMalloc((void **)&d_cell_cnt, sizeof(int) * std::max<size_t>(cell_cnt.size(), 1)));
    HIP_TRY(hipMalloc((void **)&d_mr, sizeof(int) * dev_max_run.size()));
    HIP_TRY(hipMemset(d_mr, 0, sizeof(int) * dev_max_run.size()));
    if (nwaves > 0) {
      const int wpb = TPB / WAVE;
      hipLaunchKernelGGL(tw_count_kernel, dim3((nwaves + wpb - 1) / wpb), dim3(TPB), sizeof(int) * wpb * ntiles, nullptr,
                         (const int2 *)d_wave_rows, nwaves, (const int *)D.rowptr, (const int *)D.col, tile_cols,
                         (const int *)d_map16, ntiles, d_cell_cnt, d_mr);
      HIP_TRY(hipMemcpy(cell_cnt.data(), d_cell_cnt, sizeof(int) * cell_cnt.size(), hipMemcpyDeviceToHost));
    }
    hipError_t e = hipMemcpy(dev_max_run.data(), d_mr, sizeof(int) * dev_max_run.size(), hipMemcpyDeviceToHost);
    (void)hipFree(d_mr);
    HIP_TRY(e);
  }
  auto count_cells = [&](int g, std::vector<std::vector<int>> &cnt, std::vector<int> &nsub) {
    const int w0 = g * TW_WPB, w1 = std::min(nwaves, w0 + TW_WPB);
    std::fill(nsub.begin(), nsub.end(), 0);
    for (int w = w0; w < w1; ++w) {
      std::vector<int> &c = cnt[w - w0];
      std::fill(c.begin(), c.end(), 0);
      if (on_device) for (int t = 0; t < ntiles; ++t) c[t + 1] = cell_cnt[(size_t)w * ntiles + t];
      else
      for (int k = rowptr[wave_rows[w].x]; k < rowptr[wave_rows[w].y]; ++k) c[tile_of(col[k]) + 1] += 1;
      for (int t = 0; t < ntiles; ++t) nsub[t] = std::max(nsub[t], (c[t + 1] + WIN - 1) / WIN);
      for (int t = 0; t < ntiles; ++t) c[t + 1] += c[t];   // prefix: cell start offsets
    }
  };
  // pass A: steps per workgroup (heavy tiles repeated, empty tiles skipped)
  parallel_ranges(grid, 8, [&](int g_begin, int g_end) {
    std::vector<std::vector<int>> cnt(TW_WPB, std::vector<int>((size_t)ntiles + 1));
    std::vector<int> nsub((size_t)ntiles);
    for (int g = g_begin; g < g_end; ++g) {
      count_cells(g, cnt, nsub);
      int steps = 0, touched = 0;
      for (int t = 0; t < ntiles; ++t) { steps += nsub[t]; touched += nsub[t] > 0; }
      wg_nsteps[g] = steps;
      wg_touched[g] = touched;
    }
  });
  // Locality: the share of (workgroup, tile) cells that hold entries.  1.0 for a matrix whose
  // rows scatter over all columns -- what the sweep is for.  A banded / block-local matrix
  // touches a few tiles per workgroup: its gathers are L2-local in the stream layout anyway
  // (contiguous row ranges per XCD), and the sweep only adds barriers.  Measured on MI355X
  // (profiles/r02_locality.txt), 10M x 10M, 10 per row inside a band of +-w columns.
  double touched_share = 1.0;
  {
    int64_t cells = 0;
    for (int g = 0; g < grid; ++g) cells += wg_touched[(size_t)g];
    if (grid > 0 && ntiles > 0) touched_share = (double)cells / ((double)grid * (double)ntiles);
    const char *mode_env = getenv("PDHG_SPMV");
    const bool forced = mode_env && !strcmp(mode_env, "tiled");
    // stream wins below ~0.25 (4M columns, band +-500K: share 0.245, stream 0.43 ms / sweep 0.57 ms), the
    // sweep above ~0.28 (10M columns, band +-1.5M: share 0.284, sweep 1.22 ms / stream 1.60 ms)
    const double min_share = dev_env("PDHG_TW_MIN_SHARE") ? atof(dev_env("PDHG_TW_MIN_SHARE")) : 0.26;
    D.tw_touched = touched_share;
    if (getenv("PDHG_VERBOSE"))
      fprintf(stderr, "[pdhg_hip] tiled layout %d x %d: %.3f of the (workgroup, tile) cells hold entries\n", rows, D.cols, touched_share);
    if (!forced && touched_share < min_share) return 0;
  }
  std::vector<int> wave_step_off((size_t)std::max(nwaves, 1), 0), wg_step_off((size_t)grid + 1, 0);
  std::vector<int64_t> wave_base((size_t)nwaves + 1, 0);
  int64_t step_ptr_len = 0;
  for (int g = 0; g < grid; ++g) {
    wg_step_off[g + 1] = wg_step_off[g] + wg_nsteps[g];
    for (int w = g * TW_WPB; w < std::min(nwaves, (g + 1) * TW_WPB); ++w) {
      wave_step_off[w] = (int)step_ptr_len;
      step_ptr_len += wg_nsteps[g] + 1;
      wave_base[w + 1] = wave_base[w] + (rowptr[wave_rows[w].y] - rowptr[wave_rows[w].x]);
    }
  }
  if (step_ptr_len >= INT32_MAX) return fail(-2, "tiled layout: step table too large for 32-bit offsets");
  uvec<unsigned> pk(on_device ? 0 : (size_t)wave_base[nwaves]);
  dvec tv(on_device ? 0 : (size_t)wave_base[nwaves]);
  std::vector<int> step_ptr((size_t)step_ptr_len), step_tile((size_t)wg_step_off[grid]);
  std::vector<int> max_run_of((size_t)std::max(grid, 1), 0);   // longest same-row run inside one tile, per workgroup
  // pass B: step lists, per-wave step offsets and the entries, tile-major (stable in (row, col))
  parallel_ranges(grid, 8, [&](int g_begin, int g_end) {
    std::vector<std::vector<int>> cnt(TW_WPB, std::vector<int>((size_t)ntiles + 1));
    std::vector<int> nsub((size_t)ntiles), next((size_t)ntiles);
    for (int g = g_begin; g < g_end; ++g) {
      const int w0 = g * TW_WPB, w1 = std::min(nwaves, w0 + TW_WPB);
      count_cells(g, cnt, nsub);
      int *st = step_tile.data() + wg_step_off[g];
      for (int t = 0; t < ntiles; ++t)
        for (int j = 0; j < nsub[t]; ++j) *st++ = tstart[(size_t)t];     // the tile's first column
      int max_run = 0;
      for (int w = w0; w < w1; ++w) {
        std::vector<int> &c = cnt[w - w0];
        const int r0 = wave_rows[w].x, r1 = wave_rows[w].y;
        const int64_t base = wave_base[w];
        const int total = rowptr[r1] - rowptr[r0];
        int *sp = step_ptr.data() + wave_step_off[w];
        for (int t = 0; t < ntiles; ++t) {
          const int cs = c[t], ce = c[t + 1], len = ce - cs;
          const int per = nsub[t] ? (len + nsub[t] - 1) / nsub[t] : 0;
          for (int j = 0; j < nsub[t]; ++j) *sp++ = (int)base + std::min(ce, cs + j * per);
        }
        *sp++ = (int)base + total;
        std::copy(c.begin(), c.end() - 1, next.begin());
        if (!on_device)
        for (int rr = r0; rr < r1; ++rr) {
          const unsigned rl = (unsigned)(rr - r0) << tile_shift;
          int run = 0, run_tile = -1;
          for (int k = rowptr[rr]; k < rowptr[rr + 1]; ++k) {
            const int tt = tile_of(col[k]);
            run = (tt == run_tile) ? run + 1 : 1;
            run_tile = tt;
            if (run > max_run) max_run = run;
            const int pos = next[tt]++;
            pk[(size_t)base + pos] = rl | (unsigned)(col[k] - tstart[(size_t)tt]);
            tv[(size_t)base + pos] = val[k];
          }
        }
      }
      max_run_of[g] = on_device ? dev_max_run[(size_t)g] : max_run;
    }
  });
  int max_run = 0;
  for (int g = 0; g < grid; ++g) max_run = std::max(max_run, max_run_of[g]);
  // Rows with long same-row runs inside a tile (hub rows of the PageRank LP,
  // dense-ish blocks) are summed by one lane, sequentially, to keep the
  // ascending-column order; the stream layout does that from LDS with 8 reads
  // in flight and wins on such matrices (PageRank-1M: 0.106 ms vs 0.18 ms), and
  // hubs give it natural cache locality anyway.  PDHG_SPMV=tiled overrides.
  {
    const char *mode_env = getenv("PDHG_SPMV");
    const bool forced = mode_env && !strcmp(mode_env, "tiled");
    // (Relaxed order reduces long runs with a shuffle tree -- tiled_chunk_relaxed -- but that did
    // not make the sweep competitive either: PageRank-1M 0.168 ms swept in relaxed order against
    // 0.104 ms streamed, profiles/r03_row_order.txt.  The rule stands in both modes.)
    if (!forced && max_run > 32) return 0;
  }
  D.wg_first_row.resize((size_t)grid + 1);
  for (int g = 0; g < grid; ++g) D.wg_first_row[g] = wave_rows[(size_t)g * TW_WPB].x;
  D.wg_first_row[grid] = rows;
  D.tiled = true;
  D.tw_band = touched_share < 0.9;
  D.tile_shift = tile_shift;
  D.tile_cols = tile_cols;
  D.var_tiles = !uniform;
  {
    const char *ev = dev_env("PDHG_TW_MIN_LDS_KB");
    D.tw_lds_floor = (size_t)(ev ? std::max(0, atoi(ev)) : 55) * 1024;
  }
  D.ntiles = ntiles;
  D.nwaves = nwaves;
  D.grid = grid;
  D.total_steps = (int64_t)step_tile.size();
  D.tw_entries = wave_base[nwaves];
  D.step_ptr_len = step_ptr_len;
  if (getenv("PDHG_VERBOSE")) {
    int mx = 0;
    for (int g = 0; g < grid; ++g) mx = std::max(mx, wg_nsteps[(size_t)g]);
    fprintf(stderr, "[pdhg_hip] tiled layout %d x %d: %d tiles (widest %d columns%s), %d waves x <= %d rows in %d workgroups, "
                    "steps per workgroup: mean %.1f, max %d; longest same-row run inside a tile %d\n",
            rows, D.cols, ntiles, widest, uniform ? "" : ", equal-nonzero widths", nwaves, TW_ROWS, grid,
            grid ? (double)D.total_steps / grid : 0.0, mx, max_run);
  }
  // relaxed order only where strict order is hopeless (a forced sweep over long runs); the automatically
  // chosen sweeps (runs <= 32) keep the strict-order variants and stay bit-exact with the CPU loops
  // runs of 9 ... 32: lane to lane (3; in both row orders: the sequential sums); beyond (PDHG_SPMV=tiled only, see above):
  // the shuffle tree in relaxed order (2), the LDS scratch in strict order (1 -- one workgroup per CU when rows per wave
  // are near the cap)
  D.tw_mode = max_run > 8 ? (max_run > 32 ? (relaxed ? 2 : 1) : 3) : 0;
  if (const char *ev = dev_env("PDHG_TW_MODE")) D.tw_mode = std::max(0, std::min(4, atoi(ev)));    // dev knob (4: see tune_tiled_variant)
  int rc;
  if (on_device) { D.wave_rows = d_wave_rows; d_wave_rows = nullptr; }
  else if ((rc = upload(&D.wave_rows, wave_rows))) return rc;
  if ((rc = upload(&D.wave_ent, step_ptr))) return rc;
  if ((rc = upload(&D.wave_step_off, wave_step_off))) return rc;
  if ((rc = upload(&D.step_tile, step_tile))) return rc;
  if ((rc = upload(&D.wg_step_off, wg_step_off))) return rc;
  if (on_device) {
    // the tile-major copy: a stable counting sort per wave, in HBM (tw_fill_kernel)
    const size_t total = (size_t)wave_base[nwaves];
    HIP_TRY(hipMalloc((void **)&D.pk, sizeof(unsigned) * std::max<size_t>(total, 1)));
    HIP_TRY(hipMalloc((void **)&D.tv, sizeof(double) * std::max<size_t>(total, 1)));
    if (nwaves > 0) {
      int64_t *d_base = nullptr;
      if ((rc = upload(&d_base, wave_base))) return rc;
      int tile_bits = 1;
      while ((1 << tile_bits) < ntiles) ++tile_bits;
      const int wpb = TPB / WAVE;
      hipLaunchKernelGGL(tw_fill_kernel, dim3((nwaves + wpb - 1) / wpb), dim3(TPB), sizeof(int) * wpb * ntiles, nullptr,
                         (const int2 *)D.wave_rows, nwaves, (const int64_t *)d_base, (const int *)D.rowptr, (const int *)D.col,
                         (const double *)D.val, tile_cols, (const int *)d_map16, (const int *)d_tstart, ntiles, tile_bits,
                         tile_shift, (const int *)d_cell_cnt, D.pk, D.tv);
      hipError_t e = hipDeviceSynchronize();
      (void)hipFree(d_base);
      HIP_TRY(e);
      HIP_TRY(hipGetLastError());
    }
    return 0;
  }
  if ((rc = upload(&D.pk, pk))) return rc;
  if ((rc = upload(&D.tv, tv))) return rc;
  return 0;
}

// Row blocks of one column slab: runs of consecutive rows with <= BLOCK_NNZ slab entries and
// <= MAX_ROWS_PER_BLOCK rows.  Rows that are LONG in the full matrix belong to the long-row
// path: no block may contain them (their epilogue must run exactly once, there).
// (dev) PDHG_BLOCK_CAP: a row block is closed once it holds this many entries (a single row may still bring it up to
// BLOCK_NNZ): smaller blocks = more workgroups with shorter phases.  Default: BLOCK_NNZ.
inline int block_cap() {
  const char *e = dev_env("PDHG_BLOCK_CAP");
  const int v = e ? atoi(e) : BLOCK_NNZ;
  return v < 64 ? 64 : (v > BLOCK_NNZ ? BLOCK_NNZ : v);
}

void make_row_blocks(int rows, const std::vector<int> &slab_rowptr, const std::vector<int> &full_rowptr,
                     std::vector<int2> &blks, int long_thr) {
  const int cap = block_cap();
  int r = 0;
  while (r < rows) {
    if (full_rowptr[r + 1] - full_rowptr[r] > long_thr) { ++r; continue; }
    const int r0 = r;
    int nn = 0;
    while (r < rows && (r - r0) < MAX_ROWS_PER_BLOCK && full_rowptr[r + 1] - full_rowptr[r] <= long_thr) {
      const int len = slab_rowptr[r + 1] - slab_rowptr[r];
      if (len > BLOCK_NNZ - nn || (nn > 0 && len > cap - nn)) break;
      nn += len;
      ++r;
    }
    blks.push_back(make_int2(r0, r));
  }
}

// First column of slab p of P (p = P: one past the last column).  Equal widths; PDHG_SLAB_SPLIT=f (dev knob, P = 2):
// the first slab takes the fraction f of the columns.
inline int slab_first_col(int cols, int P, int p) {
  if (p >= P) return cols;
  const int width = (cols + P - 1) / P;
  if (P == 2 && p == 1) {
    if (const char *f = dev_env("PDHG_SLAB_SPLIT")) return std::max(16, std::min(cols - 16, (int)(atof(f) * cols) / 16 * 16));
  }
  return std::min(cols, p * width);
}

// Column-slab copies for the stream layout (see spmv_stream_kernel).  Used when the
// gathered vector is 1.25 .. 4 slabs long (slab = PDHG_SLAB_MB MiB, default 4 = one XCD's
// L2); beyond that the tiled sweep is the tool.  PDHG_SLABS=0 disables.
// ... and only for rows that scatter: where the sweep was declined because a few thousand consecutive rows touch a sliver
// of the columns (banded, block-local: build_tiled's D.tw_touched), the row blocks an XCD works on at one time gather
// from a window its L2 holds anyway, and slab passes only walk the rows twice.  1M x 1M, +-5 000 columns / 100 diagonal
// blocks: 0.099 / 0.100 ms per product with two slab passes against 0.05 without (the vendor's CSR kernel: 0.048),
// profiles/r05_shape_table_1m.txt.  PDHG_SLABS=2 (dev use): slabs whatever the locality.
// Returns the number of slabs (0: none) -- ONE rule for the host and the device construction.
thread_local bool g_no_slabs = false;      // set around the builds of a group's column-chunk layouts (host_shards.hpp): they carry row sums themselves
int slab_count(const CsrDev &D, int cols) {
  if (g_no_slabs) return 0;
  const char *off = getenv("PDHG_SLABS");
  if (off && off[0] == '0') return 0;
  const char *mb = getenv("PDHG_SLAB_MB");
  const double slab_bytes = (mb ? std::max(0.25, atof(mb)) : 4.0) * 1048576.0;
  const double vec_bytes = 8.0 * (double)cols;
  if (vec_bytes <= 1.25 * slab_bytes || D.nnz < (1 << 20)) return 0;
  const int P = (int)std::ceil(vec_bytes / slab_bytes);
  if (P < 2 || P > 4) return 0;
  const bool local_rows = D.tw_touched * vec_bytes <= 0.5 * slab_bytes;
  if (local_rows && !(off && off[0] == '2')) return 0;
  return P;
}

int build_slabs(CsrDev &D, int rows, int cols, const std::vector<int> &rowptr, const ivec &col,
                const dvec &val, bool remap) {
  const int P = slab_count(D, cols);
  if (P == 0) return 0;
  const int width = (cols + P - 1) / P;
  int rc;
  for (int p = 0; p < P; ++p) {
    const int c0 = slab_first_col(cols, P, p), c1 = slab_first_col(cols, P, p + 1);
    std::vector<int> rp((size_t)rows + 1, 0);
    for (int r = 0; r < rows; ++r) {
      int cnt = 0;
      if (rowptr[r + 1] - rowptr[r] <= D.long_thr)          // long rows stay with the long-row path
        for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) cnt += (col[k] >= c0 && col[k] < c1);
      rp[(size_t)r + 1] = rp[(size_t)r] + cnt;
    }
    ivec sc((size_t)rp[(size_t)rows]);
    dvec sv((size_t)rp[(size_t)rows]);
    parallel_ranges(rows, 1 << 14, [&](int rb, int re) {
      for (int r = rb; r < re; ++r) {
        if (rowptr[r + 1] - rowptr[r] > D.long_thr) continue;
        int q = rp[(size_t)r];
        for (int k = rowptr[r]; k < rowptr[r + 1]; ++k)
          if (col[k] >= c0 && col[k] < c1) { sc[(size_t)q] = col[k]; sv[(size_t)q] = val[k]; ++q; }
      }
    });
    std::vector<int2> blks;
    make_row_blocks(rows, rp, rowptr, blks, D.long_thr);
    SlabDev S;
    S.nnz = rp[(size_t)rows];
    S.nblk = (int)blks.size();
    S.per_xcd = (S.nblk + NUM_XCD - 1) / NUM_XCD;
    S.grid = remap ? S.per_xcd * NUM_XCD : S.nblk;
    if ((rc = upload(&S.rowptr, rp))) return rc;
    if ((rc = upload(&S.col, sc))) return rc;
    if ((rc = upload(&S.val, sv))) return rc;
    if ((rc = upload(&S.blks, blks))) return rc;
    D.slabs.push_back(S);
  }
  if ((rc = alloc_zero(&D.slab_partial, rows))) return rc;
  D.grid = D.slabs.back().grid;      // the last pass runs the epilogue and writes the block partials
  return 0;
}

// The same slabs from D.rowptr / D.col / D.val in HBM (device_layout.hpp: count, scan, fill); only each slab's
// row pointers travel back, for the row blocks.  Bit-identical to build_slabs (same entries, same order, same blocks).
int build_slabs_device(CsrDev &D, int rows, int cols, const std::vector<int> &rowptr, bool remap, int P) {
  const int width = (cols + P - 1) / P;
  const int grid = (int)std::min<int64_t>(((int64_t)rows + 1 + TPB - 1) / TPB, 1 << 16);
  int rc;
  for (int p = 0; p < P; ++p) {
    const int c0 = slab_first_col(cols, P, p), c1 = slab_first_col(cols, P, p + 1);
    SlabDev S;
    HIP_TRY(hipMalloc((void **)&S.rowptr, sizeof(int) * ((size_t)rows + 1)));
    hipLaunchKernelGGL(slab_count_kernel, dim3(grid), dim3(TPB), 0, nullptr, rows, D.rowptr, D.col, c0, c1, D.long_thr, S.rowptr);
    HIP_TRY(hipGetLastError());
    if ((rc = device_exclusive_scan(S.rowptr, S.rowptr, (int64_t)rows + 1, nullptr, nullptr))) return rc;
    std::vector<int> rp((size_t)rows + 1);
    HIP_TRY(hipMemcpy(rp.data(), S.rowptr, sizeof(int) * ((size_t)rows + 1), hipMemcpyDeviceToHost));
    S.nnz = rp[(size_t)rows];
    HIP_TRY(hipMalloc((void **)&S.col, sizeof(int) * (size_t)std::max<int64_t>(S.nnz, 1)));
    HIP_TRY(hipMalloc((void **)&S.val, sizeof(double) * (size_t)std::max<int64_t>(S.nnz, 1)));
    hipLaunchKernelGGL(slab_fill_kernel, dim3(grid), dim3(TPB), 0, nullptr, rows, D.rowptr, D.col, D.val, c0, c1, D.long_thr,
                       S.rowptr, S.col, S.val);
    HIP_TRY(hipGetLastError());
    std::vector<int2> blks;
    make_row_blocks(rows, rp, rowptr, blks, D.long_thr);
    S.nblk = (int)blks.size();
    S.per_xcd = (S.nblk + NUM_XCD - 1) / NUM_XCD;
    S.grid = remap ? S.per_xcd * NUM_XCD : S.nblk;
    if ((rc = upload(&S.blks, blks))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    D.slabs.push_back(S);
  }
  if ((rc = alloc_zero(&D.slab_partial, rows))) return rc;
  D.grid = D.slabs.back().grid;      // the last pass runs the epilogue and writes the block partials
  return 0;
}

// Everything of a CsrDev that follows from the row pointers alone: row blocks, the long-row
// tables and their buffers.  (Host loops over the rows; the per-nonzero arrays are not touched.)
// Row blocks of (about) EQUAL COST for matrices whose whole product is a few workgroups per compute unit -- an
// OPT-IN (PDHG_BALANCED_BLOCKS=1), kept for the measurement it settled.  The greedy blocks above are filled to
// BLOCK_NNZ; when they number between one and four per compute unit the persistent trial kernels
// (trial_kernel.hpp: one item per workgroup and phase, four workgroups per CU) run with three full blocks on some CUs
// and four on others, and round 3's timeline of the L1-SVM LP (856 items on 256 CUs) showed every phase ending
// 3-4 us after its mean workgroup (profiles/r03_trial_kernel.txt).  The cut below makes the item count a multiple
// of the CU count and gives every block the same cost (entries + a little per row).  MEASURED (round 4,
// profiles/r04_steps_kernel.txt): the spread does not come from the block sizes -- with 1024 equal blocks the
// slowest workgroup of a phase is still 3.1-4.9 us behind the mean (12.9 against 9.8 us, 12.8 against 7.9) and the
// two extra barrier arrivals per XCD cost more than the shorter blocks save: L1-SVM 23.9k -> 23.1k it/s, random
// 100K x 100K 30.2k -> 30.6k, 250K x 250K 15.1k -> 15.5k.  Same rows, same order inside a row: the row sums keep
// their bits, and the block partials are exactly rounded double-double sums (common.hpp), which do not depend on
// the grouping.  PDHG_BALANCED_CUS overrides the CU count (tests).
inline int balanced_block_target(int greedy_blocks, int nchunks) {
  const char *ev = dev_env("PDHG_BALANCED_BLOCKS");
  if (!ev || ev[0] != '1') return 0;
  int cus = 256;
  if (const char *cv = dev_env("PDHG_BALANCED_CUS")) cus = std::max(1, atoi(cv));
  else {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  const int items = greedy_blocks + nchunks;
  if (items <= cus || items > 4 * cus) return 0;            // at most one per CU, or more than the persistent grid holds
  // blocks are launched in eights (one XCD each: the grid is per_xcd * 8), the long-row chunks ride behind them
  const int per_cu = (items + cus - 1) / cus;
  const int target = (per_cu * cus - nchunks) / NUM_XCD * NUM_XCD;
  return target > greedy_blocks ? target : 0;
}

int build_stream_tables(CsrDev &D, int rows, int cols, const std::vector<int> &rowptr, bool remap) {
  D.rows = rows;
  D.cols = cols;
  D.nnz = rowptr[rows];
  D.long_thr = long_row_threshold_from_env();
  std::vector<int2> blks;
  std::vector<int> long_row, long_chunk_ptr(1, 0), chunk_row, chunk_off, chunk_lidx;
  const int cap = block_cap();
  int r = 0;
  while (r < rows) {
    int len = rowptr[r + 1] - rowptr[r];
    D.max_row_nnz = std::max<int64_t>(D.max_row_nnz, len);
    if (len > D.long_thr) {
      const int l = (int)long_row.size();
      long_row.push_back(r);
      for (int off = 0; off < len; off += LONG_CHUNK) {
        chunk_row.push_back(r);
        chunk_off.push_back(off);
        chunk_lidx.push_back(l);
      }
      long_chunk_ptr.push_back((int)chunk_row.size());
      ++r;
      continue;
    }
    const int r0 = r;
    int nn = 0;
    while (r < rows && (r - r0) < MAX_ROWS_PER_BLOCK) {
      len = rowptr[r + 1] - rowptr[r];
      if (len > D.long_thr || len > BLOCK_NNZ - nn || (nn > 0 && len > cap - nn)) break;
      D.max_row_nnz = std::max<int64_t>(D.max_row_nnz, len);
      nn += len;
      ++r;
    }
    blks.push_back(make_int2(r0, r));
  }
  if (const int target0 = balanced_block_target((int)blks.size(), (int)chunk_row.size())) {
    // cost of a row: its entries (gathers, products) + ROW_COST for its extent, epilogue operands and stores
    constexpr int64_t ROW_COST = 2;
    auto cost_of = [&](int row) -> int64_t {
      const int len = rowptr[row + 1] - rowptr[row];
      return len > D.long_thr ? 0 : (int64_t)len + ROW_COST;
    };
    int64_t total = 0;
    for (int row = 0; row < rows; ++row) total += cost_of(row);
    std::vector<int2> cut;
    int target = target0;
    for (int attempt = 0; attempt < 4 && target > (int)blks.size(); ++attempt) {
      cut.clear();
      int64_t have = 0;               // cost of the rows placed so far
      int row = 0;
      while (row < rows) {
        if (rowptr[row + 1] - rowptr[row] > D.long_thr) { ++row; continue; }
        // this block ends where the running cost passes the next multiple of total / target (never empty; hard caps as above)
        const int64_t goal = (total * ((int64_t)cut.size() + 1) + target - 1) / target;
        const int r0 = row;
        int nn = 0;
        while (row < rows && (row - r0) < MAX_ROWS_PER_BLOCK) {
          const int len = rowptr[row + 1] - rowptr[row];
          if (len > D.long_thr || len > BLOCK_NNZ - nn) break;
          const int64_t cst = cost_of(row);
          if (row > r0 && have + cst > goal && goal - have <= cst / 2) break;    // closer to the goal without this row
          nn += len;
          have += cst;
          ++row;
          if (have >= goal) break;
        }
        cut.push_back(make_int2(r0, row));
      }
      // the hard caps can add a few blocks: the cut must stay within the multiple of the CU count it was made for
      const int excess = (int)cut.size() - target0;
      if (excess <= 0) break;
      target -= (excess + NUM_XCD - 1) / NUM_XCD * NUM_XCD;
      cut.clear();
    }
    if (getenv("PDHG_VERBOSE"))
      fprintf(stderr, "[pdhg_hip] stream layout %d x %d: %zu row blocks filled to %d entries -> %zu of equal cost (target %d)\n", rows, cols,
              blks.size(), BLOCK_NNZ, cut.size(), target0);
    if (!cut.empty()) blks.swap(cut);
  }
  D.nblk = (int)blks.size();
  D.per_xcd = (D.nblk + NUM_XCD - 1) / NUM_XCD;
  D.grid = remap ? D.per_xcd * NUM_XCD : D.nblk;
  D.nlong = (int)long_row.size();
  D.nchunks = (int)chunk_row.size();
  D.long_grid = (D.nlong + LONG_ROWS_PER_WG - 1) / LONG_ROWS_PER_WG;     // one wave per long row
  int rc;
  if ((rc = upload(&D.blks, blks))) return rc;
  if ((rc = upload(&D.long_row, long_row))) return rc;
  if ((rc = upload(&D.long_chunk_ptr, long_chunk_ptr))) return rc;
  if ((rc = upload(&D.chunk_row, chunk_row))) return rc;
  if ((rc = upload(&D.chunk_off, chunk_off))) return rc;
  if ((rc = upload(&D.chunk_lidx, chunk_lidx))) return rc;
  {
    double *z = nullptr;
    if ((rc = alloc_zero(&z, D.nlong))) return rc;
    D.long_ticket = reinterpret_cast<unsigned long long *>(z);
  }
  if ((rc = alloc_zero(&D.chunk_partial, D.nchunks))) return rc;
  return 0;
}

int build_sj_copies(CsrDev &D, int rows, const std::vector<int> &rowptr, bool remap);

int build_csr_dev(CsrDev &D, int rows, int cols, const std::vector<int> &rowptr,
                  const ivec &col, const dvec &val,
                  bool remap, int tile_cols = 0, bool relaxed = false) {
  int rc;
  if ((rc = build_stream_tables(D, rows, cols, rowptr, remap))) return rc;
  if ((rc = upload(&D.rowptr, rowptr))) return rc;
  if ((rc = upload(&D.col, col))) return rc;
  if ((rc = upload(&D.val, val))) return rc;
  if (tile_cols > 0) {
    if ((rc = build_tiled(D, rows, rowptr, &col, &val, tile_cols, relaxed))) return rc;
  }
  if (!D.tiled) {
    if ((rc = build_slabs(D, rows, cols, rowptr, col, val, remap))) return rc;
  }
  return build_sj_copies(D, rows, rowptr, remap);
}

// The same with D.rowptr / D.col / D.val ALREADY in HBM (device_layout.hpp): tables from the row
// pointers, the sweep's layout and the column slabs by the device kernels.  Cases the device mode does not
// cover (tiles of different widths, very many tiles) fetch the entries back once and take the host builders.
int build_csr_dev_resident(CsrDev &D, int rows, int cols, const std::vector<int> &rowptr, bool remap,
                           int tile_cols, bool relaxed) {
  int rc;
  if ((rc = build_stream_tables(D, rows, cols, rowptr, remap))) return rc;
  ivec col;
  dvec val;
  bool fetched = false;
  auto fetch = [&]() -> int {
    if (fetched) return 0;
    col.resize((size_t)D.nnz);
    val.resize((size_t)D.nnz);
    if (D.nnz > 0) {
      HIP_TRY(hipMemcpy(col.data(), D.col, sizeof(int) * (size_t)D.nnz, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(val.data(), D.val, sizeof(double) * (size_t)D.nnz, hipMemcpyDeviceToHost));
    }
    fetched = true;
    return 0;
  };
  if (tile_cols > 0) {
    rc = build_tiled(D, rows, rowptr, nullptr, nullptr, tile_cols, relaxed);
    if (rc == 1) {                                   // not covered on the device
      if ((rc = fetch())) return rc;
      rc = build_tiled(D, rows, rowptr, &col, &val, tile_cols, relaxed);
    }
    if (rc) return rc;
  }
  if (!D.tiled) {
    // column slabs: slab_count() -- 1.25 .. 4 slab widths of gathered vector, >= 1M nonzeros, rows that scatter
    const int P = slab_count(D, cols);
    const bool slabs = P > 0;
    if (slabs) {
      if (fetched) rc = build_slabs(D, rows, cols, rowptr, col, val, remap);
      else rc = build_slabs_device(D, rows, cols, rowptr, remap, P);
      if (rc) return rc;
    }
  }
  return build_sj_copies(D, rows, rowptr, remap);
}

// ---- sliced jagged layout (sj_kernels.hpp) ------------------------------------------------------------------------
// Host: the slot order (rows of a window by decreasing length, stable) and the slice offsets, from the row pointers
// alone; device: the entries of (d_rowptr, d_col, d_val) copied into the level-major order.  `full_rowptr` decides
// which rows are LONG (they belong to the long-row kernels: slot row -1); `rowptr` gives the lengths (a column slab's
// row pointers, or the same array).
void free_sj(SjDev &J) {
  void *ptrs[] = {J.meta, J.slice_off, J.col, J.val, J.hub};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  J = SjDev();
}

// What the builder decided for a matrix (all column slabs of it alike): the sorting window (G = 1: 256 rows, SJ_WIDE_G:
// 2 048 rows) and the hub threshold.
struct SjPlan {
  bool use = false;
  int G = 1;
  int max_len = SJ_MAX_LEN;
  double fill_narrow = 0.0, fill_wide = 0.0, hub_share = 0.0;
  double ragged_narrow = 0.0, ragged_wide = 0.0;      // share of the form's lane-levels in slices whose longest row exceeds two batches
};

int build_sj(SjDev &J, int rows, const std::vector<int> &rowptr, const std::vector<int> &full_rowptr, int long_thr,
             const int *d_rowptr, const int *d_col, const double *d_val, bool remap, int max_wgs, const SjPlan &plan) {
  const int nslices = (rows + WAVE - 1) / WAVE;
  if (nslices == 0) return 0;
  const int G = plan.G > 1 ? SJ_WIDE_G : 1, sigma = SJ_SIGMA * G, max_len = plan.max_len;
  const size_t nslots = (size_t)nslices * WAVE;
  std::vector<int> slice_off((size_t)nslices + 1, 0);
  std::vector<unsigned> meta(nslots, SJ_NONE << 16);
  const int nwin = (rows + sigma - 1) / sigma;
  std::vector<std::vector<int2>> hub_parts((size_t)std::max(1, std::min(nwin, 64)));
  std::atomic<int> part_next{0};
  parallel_ranges(nwin, 64, [&](int wb, int we) {
    std::vector<int> start((size_t)BLOCK_NNZ + 2);
    std::vector<int2> hubs;
    for (int w = wb; w < we; ++w) {
      const int r0 = w * sigma, r1 = std::min(rows, r0 + sigma);
      // stable counting sort by decreasing length; long and hub rows sort as empty ones and keep no slot row
      std::fill(start.begin(), start.end(), 0);
      auto is_long = [&](int r) { return full_rowptr[r + 1] - full_rowptr[r] > long_thr; };
      auto is_hub = [&](int r) { return !is_long(r) && rowptr[r + 1] - rowptr[r] > max_len; };
      auto len_of = [&](int r) { return (is_long(r) || is_hub(r)) ? 0 : rowptr[r + 1] - rowptr[r]; };
      for (int r = r0; r < r1; ++r) start[(size_t)(BLOCK_NNZ - len_of(r)) + 1] += 1;       // bucket = BLOCK_NNZ - length: ascending bucket = descending length
      for (int b = 0; b <= BLOCK_NNZ; ++b) start[(size_t)b + 1] += start[(size_t)b];
      for (int r = r0; r < r1; ++r) {
        const int l = len_of(r);
        const size_t slot = (size_t)r0 + (size_t)start[(size_t)(BLOCK_NNZ - l)]++;
        const bool no_slot = is_long(r) || is_hub(r);
        meta[slot] = ((no_slot ? SJ_NONE : (unsigned)(r - r0)) << 16) | (unsigned)l;
        if (is_hub(r)) hubs.push_back(make_int2(r, r + 1));
      }
    }
    if (!hubs.empty()) {      // (ranges are handed out in ascending order of wb; the parts are sorted by first row below)
      const int k = part_next.fetch_add(1);
      if (k < (int)hub_parts.size()) hub_parts[(size_t)k] = std::move(hubs);
      else { static std::mutex mu; std::lock_guard<std::mutex> lock(mu); auto &v = hub_parts.back(); v.insert(v.end(), hubs.begin(), hubs.end()); }
    }
  });
  std::vector<int2> hub;
  for (auto &v : hub_parts) hub.insert(hub.end(), v.begin(), v.end());
  std::sort(hub.begin(), hub.end(), [](const int2 &a, const int2 &b) { return a.x < b.x; });
  int64_t total = 0, hub_nnz = 0;
  for (int s = 0; s < nslices; ++s) {
    slice_off[(size_t)s] = (int)total;
    for (int q = 0; q < WAVE; ++q) total += meta[(size_t)s * WAVE + q] & 0xFFFFu;
  }
  slice_off[(size_t)nslices] = (int)total;
  for (const int2 &hb : hub) hub_nnz += rowptr[(size_t)hb.y] - rowptr[(size_t)hb.x];
  J.nslices = nslices;
  J.rows = rows;
  J.nnz = total;
  J.G = G;
  J.max_len = max_len;
  J.nhub = (int)hub.size();
  J.hub_nnz = hub_nnz;
  J.csr_rowptr = d_rowptr; J.csr_col = d_col; J.csr_val = d_val;
  J.fill_narrow = plan.fill_narrow; J.fill_wide = plan.fill_wide; J.ragged = plan.G > 1 ? plan.ragged_wide : plan.ragged_narrow;
  int cus = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  const int spw = (TPB / WAVE) * G;
  const int ngroups = (nslices + spw - 1) / spw;
  // never more workgroups than the CSR kernel's grid: every workgroup owns a block-partial slot of that grid
  J.grid = std::max(1, std::min(std::min(SJ_WGS_PER_CU * cus, ngroups), std::max(1, max_wgs)));
  if (remap) J.grid = std::max(NUM_XCD, (J.grid + NUM_XCD - 1) / NUM_XCD * NUM_XCD > max_wgs ? J.grid / NUM_XCD * NUM_XCD : (J.grid + NUM_XCD - 1) / NUM_XCD * NUM_XCD);
  int rc;
  if ((rc = upload(&J.meta, meta))) return rc;
  if ((rc = upload(&J.slice_off, slice_off))) return rc;
  if (!hub.empty() && (rc = upload(&J.hub, hub))) return rc;
  HIP_TRY(hipMalloc((void **)&J.col, sizeof(int) * (size_t)std::max<int64_t>(total, 1)));
  HIP_TRY(hipMalloc((void **)&J.val, sizeof(double) * (size_t)std::max<int64_t>(total, 1)));
  hipLaunchKernelGGL(sj_fill_kernel, dim3((nslices + TPB / WAVE - 1) / (TPB / WAVE)), dim3(TPB), 0, nullptr, nslices, spw, J.meta,
                     J.slice_off, d_rowptr, d_col, d_val, J.col, J.val);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(nullptr));
  return 0;
}

// Which stream layouts get the sliced jagged copy, and in which form: those with more row blocks than the persistent trial
// kernels take (trial_kernel.hpp: <= 1 024 items; such LPs are latency-bound and stay on the CSR row blocks) -- i.e. products
// that are bandwidth work.  Rows of more than SJ_MAX_LEN entries (the long-row path's apart) are HUB rows: the kernel runs them
// as whole-workgroup row blocks of the CSR arrays; a matrix that keeps more than 2 % of its entries there stays with
// the CSR kernels (see the end of the function).  The window: a wave's trip lasts as long as the longest row of its slice, so the FILL of a form is
// entries / sum over slices of 64 x the slice's longest row -- for the narrow form (256-row windows, one barrier per window:
// the four slices wait for the longest) 256 x the window's longest row.  The wide form (2 048-row windows, round 6) is taken
// when it fills at least 0.08 more than the narrow one (Poisson or power-law lengths; rows of one length tie and keep the
// narrow form with its operands prefetched a trip ahead); either form needs a fill >= 0.4.  Measured, round 5
// (profiles/r05_sj_layout.txt, banded 10M +-50000 / block-diagonal 10M): rows of exactly 10 entries (fill 1.0) 0.80 -> 0.58-0.61
// ms (rocSPARSE 0.70); the transposes, column counts Poisson(10) (narrow fill 0.47), 0.74-0.78 -> 0.67-0.74; round 6: DESIGN.md.
// PDHG_SJ=0 / 1 forces the layout off / on whatever the shape (tests compare the layouts bitwise), any other value means the
// automatic rule; PDHG_SJ_WIDE=0 / 1 and PDHG_SJ_MAXLEN (dev) force the form and the hub threshold.
inline SjPlan sj_plan(int nblk, int rows, const std::vector<int> &rowptr, const std::vector<int> &full_rowptr, int long_thr) {
  SjPlan P;
  if (const char *ev = dev_env("PDHG_SJ_MAXLEN")) P.max_len = std::max(1, std::min(atoi(ev), std::min(long_thr, BLOCK_NNZ)));
  const char *ev = getenv("PDHG_SJ");
  const int force = (ev && ev[0] == '0' && !ev[1]) ? 0 : ((ev && ev[0] == '1' && !ev[1]) ? 1 : -1);
  if (force == 0) return P;
  if (force < 0 && nblk <= 1024) return P;
  const int wide_sigma = SJ_SIGMA * SJ_WIDE_G;
  const int nwin = (rows + wide_sigma - 1) / wide_sigma;
  constexpr int NACC = 6;
  std::vector<int64_t> acc((size_t)NACC * std::max(1, std::min(nwin, 64)), 0);  // per range: entries, narrow capacity, wide capacity, hub entries, ragged narrow / wide capacity
  std::atomic<int> part_next{0};
  parallel_ranges(nwin, 64, [&](int wb, int we) {
    int64_t entries = 0, cap_narrow = 0, cap_wide = 0, hub_entries = 0, rag_narrow = 0, rag_wide = 0;
    std::vector<int> hist((size_t)P.max_len + 1);
    for (int w = wb; w < we; ++w) {
      const int r0 = w * wide_sigma, r1 = std::min(rows, r0 + wide_sigma);
      std::fill(hist.begin(), hist.end(), 0);
      for (int g0 = r0; g0 < r1; g0 += SJ_SIGMA) {
        int longest = 0;
        for (int r = g0; r < std::min(r1, g0 + SJ_SIGMA); ++r) {
          if (full_rowptr[r + 1] - full_rowptr[r] > long_thr) continue;
          const int l = rowptr[r + 1] - rowptr[r];
          if (l > P.max_len) { hub_entries += l; continue; }
          hist[(size_t)l] += 1;
          longest = std::max(longest, l);
          entries += l;
        }
        cap_narrow += (int64_t)SJ_SIGMA * longest;
        if (longest > 2 * SJ_U) rag_narrow += (int64_t)SJ_SIGMA * longest;
      }
      // the window's rows by decreasing length: the longest row of every 64
      int seen = 0, next_head = 0;
      for (int l = P.max_len; l >= 1; --l) {
        const int c = hist[(size_t)l];
        while (next_head < seen + c) { cap_wide += (int64_t)WAVE * l; if (l > 2 * SJ_U) rag_wide += (int64_t)WAVE * l; next_head += WAVE; }
        seen += c;
      }
    }
    const int k = std::min(part_next.fetch_add(1), (int)(acc.size() / NACC) - 1);
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    int64_t *a = &acc[(size_t)NACC * k];
    a[0] += entries; a[1] += cap_narrow; a[2] += cap_wide; a[3] += hub_entries; a[4] += rag_narrow; a[5] += rag_wide;
  });
  int64_t entries = 0, cap_narrow = 0, cap_wide = 0, hub_entries = 0, rag_narrow = 0, rag_wide = 0;
  for (size_t k = 0; k < acc.size() / NACC; ++k) {
    const int64_t *a = &acc[NACC * k];
    entries += a[0]; cap_narrow += a[1]; cap_wide += a[2]; hub_entries += a[3]; rag_narrow += a[4]; rag_wide += a[5];
  }
  P.ragged_narrow = cap_narrow > 0 ? (double)rag_narrow / (double)cap_narrow : 0.0;
  P.ragged_wide = cap_wide > 0 ? (double)rag_wide / (double)cap_wide : 0.0;
  P.fill_narrow = cap_narrow > 0 ? (double)entries / (double)cap_narrow : 0.0;
  P.fill_wide = cap_wide > 0 ? (double)entries / (double)cap_wide : 0.0;
  P.hub_share = entries + hub_entries > 0 ? (double)hub_entries / (double)(entries + hub_entries) : 0.0;
  P.G = P.fill_wide >= P.fill_narrow + 0.08 ? SJ_WIDE_G : 1;
  if (const char *wv = dev_env("PDHG_SJ_WIDE")) P.G = wv[0] != '0' ? SJ_WIDE_G : 1;
  if (force == 1) { P.use = true; return P; }
  // ... and not a power-law body: a slice is walked in batches of 16 levels, one dependent round trip each whatever the
  // number of lanes still active, so slices of 30 ... 128 levels with a handful of long rows run at the latency of their
  // batches (PageRank-1M on this layout, hub rows split off: 0.22-0.37 ms per product against 0.10 on the CSR row blocks,
  // profiles/r06_sj_wide.txt).  Such matrices keep the row blocks: at most a tenth of the lane-levels may sit in slices
  // beyond two batches, at most 2 % of the entries in hub rows.
  P.use = entries > 0 && P.hub_share <= 0.02 && (P.G > 1 ? P.fill_wide : P.fill_narrow) >= 0.4 &&
          (P.G > 1 ? P.ragged_wide : P.ragged_narrow) <= 0.10;
  return P;
}

// The persistent software-pipelined form of the CSR row blocks (spmv_stream_pipe_kernel): the blocks' extent words and the
// grid.  For the same products as the sliced jagged copy -- more row blocks than the persistent trial kernels take --
// whatever the row lengths; PDHG_STREAM_PIPE=0 / 1 (dev) forces it off / on.  `rowptr`: the row pointers `blks` was cut from.
// Row blocks of very different cost (a hub row fills a block by itself and is added by one lane or one wave) want the
// hardware's dynamic dispatch of the plain kernel, not a fixed walk: PageRank-1M lost 6 % on the pipelined launch (0.105
// against 0.099 ms per product) where banded / block-diagonal matrices with Poisson(10) rows gained 5-6 % (0.77 -> 0.73 ms;
// profiles/r05_sj_layout.txt).  So: only matrices whose rows (the long-row path's apart) stay within SJ_MAX_LEN entries.
// Round 6: ALSO every matrix of at most 1 024 row blocks, whatever its rows.  The persistent grid (4 workgroups per CU) then
// holds one workgroup PER BLOCK -- no walk, the hardware still deals the blocks -- and what the kernel buys is its request
// order: the extent word gives a block's entries, row extents and epilogue operands in ONE round trip, where the plain
// kernel chains block table -> row pointers -> entries -> gathers -> row extents -> operands (six).  These launches are
// one resident wave of workgroups, i.e. pure latency: the products of a termination / restart check, the event brackets
// behind the roofline figures and the graph-path trials of such LPs (the persistent trial kernels walk the row blocks
// themselves).  L1-SVM LP, products as separate kernels: profiles/r06_stream_waitcnt.txt.
inline bool stream_pipe_wanted(int nblk, int64_t max_row_nnz, int long_thr) {
  if (const char *ev = dev_env("PDHG_STREAM_PIPE")) return ev[0] != '0';
  if (nblk <= 1024) return nblk >= 2 * NUM_XCD;
  return std::min<int64_t>(max_row_nnz, long_thr) <= SJ_MAX_LEN;
}
// want_pipe == false: the extent words alone (round 6: the plain kernel reads a block's rows AND entry range from one word
// instead of the block table, then two row pointers), the persistent kernel's grid stays 0.
int build_block_extents(int4 **ext_out, int *grid_out, const int2 *d_blks, int nblk, int max_wgs, const std::vector<int> &rowptr, bool remap,
                        bool want_pipe = true) {
  if (nblk <= 0) return 0;
  std::vector<int2> blks((size_t)nblk);
  HIP_TRY(hipMemcpy(blks.data(), d_blks, sizeof(int2) * (size_t)nblk, hipMemcpyDeviceToHost));
  std::vector<int4> ext((size_t)nblk);
  for (int b = 0; b < nblk; ++b) ext[(size_t)b] = make_int4(blks[(size_t)b].x, blks[(size_t)b].y, rowptr[(size_t)blks[(size_t)b].x], rowptr[(size_t)blks[(size_t)b].y]);
  int cus = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
  }
  // never more workgroups than the plain kernel's grid: every workgroup owns a block-partial slot of that grid
  int grid = std::max(1, std::min(std::min(STREAM_PIPE_WGS_PER_CU * cus, nblk), std::max(1, max_wgs)));
  if (remap) grid = std::max(NUM_XCD, (grid + NUM_XCD - 1) / NUM_XCD * NUM_XCD > max_wgs ? grid / NUM_XCD * NUM_XCD : (grid + NUM_XCD - 1) / NUM_XCD * NUM_XCD);
  if (want_pipe) *grid_out = grid;
  return upload(ext_out, ext);
}

// after the stream tables / slabs exist and D.rowptr / D.col / D.val are in HBM
int build_sj_copies_only(CsrDev &D, int rows, const std::vector<int> &rowptr, bool remap);
int build_sj_copies(CsrDev &D, int rows, const std::vector<int> &rowptr, bool remap) {
  int rc = build_sj_copies_only(D, rows, rowptr, remap);
  if (rc || D.tiled || !D.segs.empty()) return rc;
  // products that did not get the sliced jagged copy: the pipelined launch of the row blocks
  // (every row-block layout gets its extent words: the plain kernel reads them too; the persistent grid only where wanted)
  if (!D.slabs.empty()) {
    if (D.slabs.front().sj.on()) return 0;
    int nblk_max = 0;
    for (const SlabDev &S : D.slabs) nblk_max = std::max(nblk_max, S.nblk);
    const bool pipe = stream_pipe_wanted(nblk_max, D.max_row_nnz, D.long_thr);
    for (SlabDev &S : D.slabs) {
      std::vector<int> rp((size_t)rows + 1);
      HIP_TRY(hipMemcpy(rp.data(), S.rowptr, sizeof(int) * ((size_t)rows + 1), hipMemcpyDeviceToHost));
      if ((rc = build_block_extents(&S.ext, &S.pipe_grid, S.blks, S.nblk, S.grid, rp, remap, pipe))) return rc;
    }
    return 0;
  }
  if (D.sj.on() || D.grid <= 0) return 0;
  return build_block_extents(&D.ext, &D.pipe_grid, D.blks, D.nblk, D.grid, rowptr, remap, stream_pipe_wanted(D.nblk, D.max_row_nnz, D.long_thr));
}
int build_sj_copies_only(CsrDev &D, int rows, const std::vector<int> &rowptr, bool remap) {
  if (D.tiled || !D.segs.empty()) return 0;
  int rc;
  if (!D.slabs.empty()) {
    // all slabs or none (the passes hand the row sums on in one format either way; one rule keeps the kernel names simple)
    int nblk_max = 0;
    for (const SlabDev &S : D.slabs) nblk_max = std::max(nblk_max, S.nblk);
    const SjPlan plan = sj_plan(nblk_max, rows, rowptr, rowptr, D.long_thr);
    if (!plan.use) return 0;
    for (SlabDev &S : D.slabs) {
      std::vector<int> rp((size_t)rows + 1);
      HIP_TRY(hipMemcpy(rp.data(), S.rowptr, sizeof(int) * ((size_t)rows + 1), hipMemcpyDeviceToHost));
      if ((rc = build_sj(S.sj, rows, rp, rowptr, D.long_thr, S.rowptr, S.col, S.val, remap, S.grid, plan))) return rc;
    }
    return 0;
  }
  if (D.grid <= 0) return 0;
  const SjPlan plan = sj_plan(D.nblk, rows, rowptr, rowptr, D.long_thr);
  if (!plan.use) return 0;
  return build_sj(D.sj, rows, rowptr, rowptr, D.long_thr, D.rowptr, D.col, D.val, remap, D.grid, plan);
}

void free_csr_dev(CsrDev &D) {
  for (CsrDev &S : D.segs) free_csr_dev(S);
  void *ptrs[] = {D.rowptr, D.col, D.val, D.blks, D.long_row, D.long_chunk_ptr,
                  D.chunk_row, D.chunk_off, D.chunk_lidx, D.long_ticket, D.chunk_partial, D.wave_rows, D.wave_ent, D.pk, D.tv,
                  D.wave_step_off, D.step_tile, D.wg_step_off};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  for (SlabDev &S : D.slabs) {
    void *sp[] = {S.rowptr, S.col, S.val, S.blks, S.ext};
    for (void *p : sp) if (p) (void)hipFree(p);
    free_sj(S.sj);
  }
  free_sj(D.sj);
  if (D.ext) (void)hipFree(D.ext);
  if (D.slab_partial) (void)hipFree(D.slab_partial);
  D = CsrDev();
}

}  // namespace
