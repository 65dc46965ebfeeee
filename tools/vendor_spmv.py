#!/usr/bin/env python3
"""rocSPARSE's CSR SpMV timed on the matrices the product kernels run on: the INDEPENDENT comparator of the roofline
claims (measurement aid; the package never imports this).  ``time_csr(A_csr, x)`` -> {alg: ms, ..., "best_ms", "best_alg"}.

    python tools/vendor_spmv.py [--shape random|pagerank|l1svm|banded|blockdiag] [--n N]
prints the vendor's times for A x and A' y beside the product kernels' (HIP events around pdhg_spmv / pdhg_spmv_t)."""
import ctypes
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "vendor_spmv.cpp")
LIB = os.path.join(HERE, "libvendor_spmv.so")
ALGS = {"adaptive": 2, "rowsplit": 3, "lrb": 7, "nnzsplit": 8}


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, "-O2", "-fPIC", "-shared", "-std=c++17", SRC, "-lrocsparse", "-o", LIB + ".tmp"], check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


_L = None


def _lib():
    global _L
    if _L is None:
        _L = ctypes.CDLL(build())
        i64, vp = ctypes.c_int64, ctypes.c_void_p
        _L.vendor_spmv_time.argtypes = [i64, i64, i64, vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, vp, vp]
    return _L


def time_csr(A, x=None, reps=20, algs=("adaptive", "rowsplit", "lrb", "nnzsplit"), check=True):
    """A: scipy CSR (int32-indexable).  Returns ms per product for every algorithm rocSPARSE accepts, preprocessing apart."""
    A = A.tocsr()
    A.sort_indices()
    rows, cols = A.shape
    rp = np.ascontiguousarray(A.indptr, dtype=np.int32)
    ci = np.ascontiguousarray(A.indices, dtype=np.int32)
    va = np.ascontiguousarray(A.data, dtype=np.float64)
    if x is None:
        x = np.random.default_rng(1).standard_normal(cols)
    x = np.ascontiguousarray(x, dtype=np.float64)
    ids = np.array([ALGS[a] for a in algs], dtype=np.int32)
    out = np.zeros(2 * len(algs))
    y = np.zeros(rows)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)     # noqa: E731
    rc = _lib().vendor_spmv_time(rows, cols, A.nnz, p(rp), p(ci), p(va), p(x), p(ids), len(algs), reps, p(out), p(y))
    if rc:
        raise RuntimeError(f"vendor_spmv_time failed: {rc}")
    res = {"preprocess_ms": {}}
    for k, a in enumerate(algs):
        if out[2 * k] >= 0:
            res[a] = round(float(out[2 * k]), 5)
            res["preprocess_ms"][a] = round(float(out[2 * k + 1]), 3)
    ok = {a: res[a] for a in algs if a in res}
    if ok:
        res["best_alg"] = min(ok, key=ok.get)
        res["best_ms"] = ok[res["best_alg"]]
    if check and ok:
        want = A @ x
        res["max_rel_err"] = float(np.max(np.abs(y - want)) / max(1e-300, np.max(np.abs(want))))
    return res


def main():
    import argparse
    sys.path.insert(0, ROOT)
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="random")
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--nnz-per-row", type=int, default=10)
    args = ap.parse_args()
    from tools.shape_table import SHAPES, make_shape, product_ms
    named = {"banded50k": ("banded", dict(band=50_000)), "blockdiag": ("blockdiag", {}), "configS": ("random", {})}
    if args.shape in named:
        p = make_shape(named[args.shape][0], **named[args.shape][1])
    else:
        p = make_shape(args.shape, m=args.n, n=args.n, k=args.nnz_per_row)
    A = p.constraint_matrix.tocsr()
    print(f"{args.shape} {A.shape} nnz={A.nnz}")
    print("  vendor A x  :", time_csr(A))
    print("  vendor A' y :", time_csr(A.T.tocsr()))
    print("  product     :", product_ms(p))


if __name__ == "__main__":
    main()
