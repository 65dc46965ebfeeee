// vendor_spmv.cpp -- MEASUREMENT AID, not product code: nothing under firstorderlp.jl_amd/ loads it.
//
// The vendor's CSR SpMV (rocSPARSE, every CSR algorithm it offers) timed on the same device and the same
// CSR(A) / CSR(A') arrays the product kernels consume, as the INDEPENDENT comparator for the roofline
// claims of DESIGN.md section 4 (SURVEY.md line 16 allows rocSPARSE as a cross-check, never on the product
// path).  bench.py quotes the best algorithm's time as roofline.vendor_spmv_ms; tools/shape_table.py puts
// it beside every product kernel.  Built on demand by tools/vendor_spmv.py:
//   hipcc -O2 -fPIC -shared tools/vendor_spmv.cpp -lrocsparse -o tools/libvendor_spmv.so
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#pragma clang diagnostic ignored "-Wdeprecated-declarations"

#define HIP_CK(e)                                                                           \
  do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "[vendor_spmv] %s: %s\n", #e, hipGetErrorString(_e)); return 1000 + (int)_e; } } while (0)
#define RS_CK(e)                                                                            \
  do { rocsparse_status _s = (e); if (_s != rocsparse_status_success) { fprintf(stderr, "[vendor_spmv] %s: status %d\n", #e, (int)_s); return 2000 + (int)_s; } } while (0)

extern "C" {

// y = A x with CSR (rows x cols, 0-based int32 indices, host arrays).  algs[k] in {2 adaptive, 3 rowsplit ("stream"), 7 lrb,
// 8 nnzsplit}; for each: out_ms[2k] = average kernel-side ms over `reps` back-to-back products (HIP events on the handle's
// stream), out_ms[2k+1] = preprocessing ms (once).  y_out (host, rows doubles or NULL) receives the LAST algorithm's result.
// A negative out_ms[2k] means the algorithm was refused.
int vendor_spmv_time(int64_t rows, int64_t cols, int64_t nnz, const int *rowptr, const int *col, const double *val, const double *x,
                     const int *algs, int nalgs, int reps, double *out_ms, double *y_out) {
  int *d_rp = nullptr, *d_ci = nullptr;
  double *d_v = nullptr, *d_x = nullptr, *d_y = nullptr;
  HIP_CK(hipMalloc(&d_rp, sizeof(int) * (size_t)(rows + 1)));
  HIP_CK(hipMalloc(&d_ci, sizeof(int) * (size_t)std::max<int64_t>(nnz, 1)));
  HIP_CK(hipMalloc(&d_v, sizeof(double) * (size_t)std::max<int64_t>(nnz, 1)));
  HIP_CK(hipMalloc(&d_x, sizeof(double) * (size_t)cols));
  HIP_CK(hipMalloc(&d_y, sizeof(double) * (size_t)rows));
  HIP_CK(hipMemcpy(d_rp, rowptr, sizeof(int) * (size_t)(rows + 1), hipMemcpyHostToDevice));
  HIP_CK(hipMemcpy(d_ci, col, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
  HIP_CK(hipMemcpy(d_v, val, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice));
  HIP_CK(hipMemcpy(d_x, x, sizeof(double) * (size_t)cols, hipMemcpyHostToDevice));
  hipStream_t stream;
  HIP_CK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  rocsparse_handle h;
  RS_CK(rocsparse_create_handle(&h));
  RS_CK(rocsparse_set_stream(h, stream));
  rocsparse_spmat_descr A;
  rocsparse_dnvec_descr X, Y;
  RS_CK(rocsparse_create_csr_descr(&A, rows, cols, nnz, d_rp, d_ci, d_v, rocsparse_indextype_i32, rocsparse_indextype_i32,
                                   rocsparse_index_base_zero, rocsparse_datatype_f64_r));
  RS_CK(rocsparse_create_dnvec_descr(&X, cols, d_x, rocsparse_datatype_f64_r));
  RS_CK(rocsparse_create_dnvec_descr(&Y, rows, d_y, rocsparse_datatype_f64_r));
  hipEvent_t e0, e1;
  HIP_CK(hipEventCreate(&e0));
  HIP_CK(hipEventCreate(&e1));
  const double alpha = 1.0, beta = 0.0;
  for (int k = 0; k < nalgs; ++k) {
    out_ms[2 * k] = -1.0;
    out_ms[2 * k + 1] = 0.0;
    const rocsparse_spmv_alg alg = (rocsparse_spmv_alg)algs[k];
    size_t bs = 0;
    if (rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, X, &beta, Y, rocsparse_datatype_f64_r, alg, rocsparse_spmv_stage_buffer_size,
                       &bs, nullptr) != rocsparse_status_success) continue;
    void *buf = nullptr;
    HIP_CK(hipMalloc(&buf, std::max<size_t>(bs, 16)));
    HIP_CK(hipEventRecord(e0, stream));
    rocsparse_status st = rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, X, &beta, Y, rocsparse_datatype_f64_r, alg,
                                         rocsparse_spmv_stage_preprocess, &bs, buf);
    HIP_CK(hipEventRecord(e1, stream));
    HIP_CK(hipStreamSynchronize(stream));
    if (st != rocsparse_status_success) { (void)hipFree(buf); continue; }
    float pre = 0.f;
    HIP_CK(hipEventElapsedTime(&pre, e0, e1));
    out_ms[2 * k + 1] = pre;
    bool ok = true;
    for (int w = 0; w < 3 && ok; ++w)
      ok = rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, X, &beta, Y, rocsparse_datatype_f64_r, alg, rocsparse_spmv_stage_compute, &bs,
                          buf) == rocsparse_status_success;
    HIP_CK(hipStreamSynchronize(stream));
    if (ok) {
      HIP_CK(hipEventRecord(e0, stream));
      for (int r = 0; r < reps && ok; ++r)
        ok = rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, X, &beta, Y, rocsparse_datatype_f64_r, alg, rocsparse_spmv_stage_compute,
                            &bs, buf) == rocsparse_status_success;
      HIP_CK(hipEventRecord(e1, stream));
      HIP_CK(hipStreamSynchronize(stream));
      float ms = 0.f;
      HIP_CK(hipEventElapsedTime(&ms, e0, e1));
      if (ok) out_ms[2 * k] = ms / std::max(reps, 1);
    }
    (void)hipFree(buf);
    // the preprocessing result lives in the matrix descriptor: a fresh one for the next algorithm
    RS_CK(rocsparse_destroy_spmat_descr(A));
    RS_CK(rocsparse_create_csr_descr(&A, rows, cols, nnz, d_rp, d_ci, d_v, rocsparse_indextype_i32, rocsparse_indextype_i32,
                                     rocsparse_index_base_zero, rocsparse_datatype_f64_r));
  }
  if (y_out) HIP_CK(hipMemcpy(y_out, d_y, sizeof(double) * (size_t)rows, hipMemcpyDeviceToHost));
  (void)rocsparse_destroy_spmat_descr(A);
  (void)rocsparse_destroy_dnvec_descr(X);
  (void)rocsparse_destroy_dnvec_descr(Y);
  (void)rocsparse_destroy_handle(h);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipStreamDestroy(stream);
  (void)hipFree(d_rp); (void)hipFree(d_ci); (void)hipFree(d_v); (void)hipFree(d_x); (void)hipFree(d_y);
  return 0;
}

}  // extern "C"
