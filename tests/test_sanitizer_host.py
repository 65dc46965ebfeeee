"""The C-ABI library's HOST side under AddressSanitizer + UndefinedBehaviorSanitizer (no GPU needed): the host-only
entry points -- the nnz-balanced row partition of csrc/dist.hpp, argument validation of the creators -- are driven
through libpdhg_hip_asan.so (`_lib.build_sanitized("asan")`: same translation unit, -fsanitize=address,undefined,
kernels compiled as usual) in a child process that preloads the ASan runtime.  Any report aborts the child."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = textwrap.dedent(r"""
    import ctypes, sys
    import numpy as np
    import scipy.sparse as sp
    L = ctypes.CDLL(sys.argv[1])
    L.pdhg_last_error.restype = ctypes.c_char_p
    i64p = ctypes.POINTER(ctypes.c_int64)
    dp = ctypes.POINTER(ctypes.c_double)
    pi = lambda a: a.ctypes.data_as(i64p)
    pd = lambda a: a.ctypes.data_as(dp)
    assert L.pdhg_abi_version() == int(sys.argv[2])
    L.pdhg_partition_rows.argtypes = [ctypes.c_int64, ctypes.c_int64, i64p, i64p, ctypes.c_int, ctypes.c_int, i64p]
    for (m, n, dens, world, base) in [(500, 300, 0.02, 4, 0), (37, 1000, 0.2, 3, 1), (1, 5, 1.0, 1, 0), (2000, 3, 0.5, 16, 1),
                                      (50, 40, 0.0, 4, 0)]:
        A = sp.random(m, n, density=dens, random_state=m + n, format="csc")
        colptr = A.indptr.astype(np.int64) + base
        rowval = A.indices.astype(np.int64) + base
        out = np.full(world + 1, -7, dtype=np.int64)
        rc = L.pdhg_partition_rows(m, n, pi(colptr), pi(rowval) if len(rowval) else pi(np.zeros(1, dtype=np.int64)), base, world, pi(out))
        assert rc == 0, L.pdhg_last_error()
        assert out[0] == 0 and out[-1] == m and np.all(np.diff(out) >= 0), out
        # the shards' nonzero counts differ by at most the heaviest row
        rows = np.bincount(A.indices, minlength=m)
        per = [rows[out[k]:out[k + 1]].sum() for k in range(world)]
        assert max(per) - min(per) <= 2 * max(rows.max(), 1) + A.nnz // world, per
        # row indices outside the matrix must not be used as indices (they are skipped by the counting pass)
        if len(rowval) > 3:
            bad = rowval.copy(); bad[::3] = 10 ** 12; bad[1::3] = -5
            assert L.pdhg_partition_rows(m, n, pi(colptr), pi(bad), base, world, pi(out)) == 0
    # argument validation: every one of these must come back with an error code, not touch memory
    out = np.zeros(8, dtype=np.int64)
    assert L.pdhg_partition_rows(10, 10, None, None, 0, 2, pi(out)) != 0
    cp = np.zeros(11, dtype=np.int64)
    assert L.pdhg_partition_rows(10, 10, pi(cp), None, 0, 0, pi(out)) != 0          # world < 1
    assert L.pdhg_partition_rows(-1, 10, pi(cp), None, 0, 2, pi(out)) != 0
    cp1 = cp + 1
    assert L.pdhg_partition_rows(10, 10, pi(cp1), None, 0, 2, pi(out)) != 0         # colptr[0] != index_base
    h = ctypes.c_void_p()
    L.pdhg_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, i64p, i64p, dp,
                              ctypes.c_int, dp, dp, dp, dp, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
    z = np.zeros(4)
    one = np.zeros(3, dtype=np.int64)
    assert L.pdhg_create(None, 2, 2, 0, pi(one), pi(one), pd(z), 0, pd(z), pd(z), pd(z), pd(z), 0, 0, None) != 0
    assert L.pdhg_create(ctypes.byref(h), 2, 2, 0, None, None, None, 0, None, None, None, None, 0, 0, None) != 0
    assert L.pdhg_create(ctypes.byref(h), -2, 2, 0, pi(one), pi(one), pd(z), 0, pd(z), pd(z), pd(z), pd(z), 0, 0, None) != 0
    assert L.pdhg_create(ctypes.byref(h), 2, 2, 0, pi(one), pi(one), pd(z), 7, pd(z), pd(z), pd(z), pd(z), 0, 0, None) != 0
    assert L.pdhg_create(ctypes.byref(h), 2, 2, 0, pi(one), pi(one), pd(z), 0, pd(z), pd(z), pd(z), pd(z), 5, 0, None) != 0   # more equalities than rows
    # the >2^31-nonzero route (limit lowered): null arrays and out-of-range row indices are refused before any indexing
    import os
    os.environ["PDHG_MAX_SHARD_NNZ"] = "4"
    A = sp.random(40, 30, density=0.3, random_state=3, format="csc")
    colptr, rowval, nz = A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.copy()
    v = np.zeros(40)
    assert L.pdhg_create(ctypes.byref(h), 40, 30, A.nnz, pi(colptr), pi(rowval), None, 0, pd(v), pd(v), pd(v), pd(v), 0, 0, None) != 0
    bad = rowval.copy(); bad[5] = 40
    assert L.pdhg_create(ctypes.byref(h), 40, 30, A.nnz, pi(colptr), pi(bad), pd(nz), 0, pd(v), pd(v), pd(v), pd(v), 0, 0, None) != 0
    assert b"row index" in L.pdhg_last_error()
    print("SANITIZED_HOST_OK")
""")


def test_host_entry_points_under_asan_and_ubsan(tmp_path):
    import shutil
    from firstorderlp_jl_amd import _lib
    if not shutil.which(os.environ.get("HIPCC", "hipcc")):
        pytest.skip("no hipcc on this box: the sanitized library cannot be built")
    rt = _lib.sanitizer_runtime("asan")
    if not rt:
        pytest.skip("clang's ASan runtime (libclang_rt.asan) is not in this toolchain")
    try:
        path = _lib.build_sanitized("asan")
    except Exception as exc:      # the sanitized copy is test infrastructure: its build never gates the product's
        pytest.skip(f"the sanitized library does not build here: {exc}")
    script = tmp_path / "drive.py"
    script.write_text(DRIVER)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    env.pop("PDHG_HIP_LIB", None)
    r = subprocess.run([sys.executable, str(script), path, str(_lib.ABI_VERSION)], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "SANITIZED_HOST_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-4000:]
