"""Error behaviour of the C ABI: every entry point returns an int (0 ok, <0 bad
argument / unsupported), no exception or crash crosses the boundary; the Python
binding turns non-zero into PdhgHipError carrying pdhg_last_error()."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp

from firstorderlp_jl_amd import HipPdhgEngine, _lib, linear_programming_problem
from tests import helpers as H

pytestmark = pytest.mark.gpu
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int64)


def _raw_create(m, n, colptr, rowval, nzval, base, ne, c=None, b=None):
    L = _lib.lib()
    colptr = np.ascontiguousarray(colptr, dtype=np.int64)
    rowval = np.ascontiguousarray(rowval, dtype=np.int64)
    nzval = np.ascontiguousarray(nzval, dtype=np.float64)
    c = np.zeros(n) if c is None else c
    b = np.zeros(m) if b is None else b
    lb, ub = np.zeros(n), np.ones(n)
    h = ctypes.c_void_p()
    rc = L.pdhg_create(ctypes.byref(h), m, n, len(nzval), colptr.ctypes.data_as(_ip),
                       rowval.ctypes.data_as(_ip), nzval.ctypes.data_as(_dp), base,
                       c.ctypes.data_as(_dp), b.ctypes.data_as(_dp), lb.ctypes.data_as(_dp),
                       ub.ctypes.data_as(_dp), ne, -1, None)
    if rc == 0:
        L.pdhg_destroy(h)
    return rc, L.pdhg_last_error().decode()


def test_create_rejects_bad_arguments(gpu_required):
    ok = ([0, 1, 2], [0, 1], [1.0, 2.0])
    assert _raw_create(2, 2, *ok, 0, 1)[0] == 0
    assert _raw_create(2, 2, [1, 2, 3], [1, 2], [1.0, 2.0], 1, 1)[0] == 0     # Julia's 1-based arrays
    rc, msg = _raw_create(2, 2, *ok, 2, 1)
    assert rc < 0 and "index_base" in msg
    rc, msg = _raw_create(2, 2, *ok, 0, 3)
    assert rc < 0 and "num_equalities" in msg
    rc, msg = _raw_create(2, 2, [0, 1, 2], [0, 5], [1.0, 2.0], 0, 1)
    assert rc < 0 and "rowval" in msg
    rc, msg = _raw_create(2, 2, [0, 2, 1], [0, 1], [1.0, 2.0], 0, 1)
    assert rc < 0
    rc, msg = _raw_create(2, 2, [1, 2, 3], [0, 1], [1.0, 2.0], 0, 1)
    assert rc < 0 and "colptr" in msg


def test_unsupported_and_misordered_calls(gpu_required):
    eng = HipPdhgEngine.from_problem(H.example_lp())
    with pytest.raises(_lib.PdhgHipError, match="average is empty"):
        eng.restart_to_average()
    with pytest.raises(_lib.PdhgHipError, match="pdhg_set_original_problem"):
        eng.eval_point(_lib.POINT_CURRENT)
    with pytest.raises(_lib.PdhgHipError, match="rank out of range"):
        HipPdhgEngine.from_problem(H.example_lp(), unique_id=b"\0" * 128, rank=3, world=2)
    with pytest.raises(_lib.PdhgHipError, match="device_id out of range"):
        HipPdhgEngine.from_problem(H.example_lp(), device_ids=[0, 99])
    with pytest.raises(_lib.PdhgHipError, match="range"):
        eng.trust_region_bound(_lib.POINT_CURRENT, 1.0, 1.0, 1.0, 7)
    with pytest.raises(_lib.PdhgHipError, match="alpha"):
        eng.rescale(0, False, 2.5)
    # the engine is still usable after errors
    raw = eng.trial_step(0.1, 1.0, 1.0)
    assert np.all(np.isfinite(raw))


def test_roctx_ranges_bind_at_run_time(gpu_required, tmp_path):
    """PDHG_ROCTX=1: the entry points and the fused products push / pop named roctx ranges through a marker library
    bound with dlopen (no link-time dependency).  A child process takes a few steps with the ranges on; under rocprofv3's
    marker trace (when the tool is on the box) the range names must show up in its output."""
    import os
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np, folp_loader\n"
            "pkg = folp_loader.load()\n"
            "from firstorderlp_jl_amd.generators import random_lp\n"
            "p = random_lp(3000, 2500, 6, seed=1)\n"
            "e = pkg.HipPdhgEngine.from_problem(p)\n"
            "for _ in range(3):\n"
            "    raw = e.trial_step(0.1, 1.0, 1.0); e.accept(0.1)\n"
            "print('ROCTX_CHILD_OK', float(raw[1]))\n" % root)
    script = tmp_path / "child.py"
    script.write_text(code)
    env = dict(os.environ, PDHG_ROCTX="1", PDHG_GRAPH="0", TMPDIR="/tmp")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ROCTX_CHILD_OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
    assert "no roctx library could be bound" not in r.stderr, r.stderr[-500:]
    rp = shutil.which("rocprofv3")
    if rp:
        out = tmp_path / "trace"
        r = subprocess.run([rp, "--marker-trace", "--kernel-trace", "--output-format", "csv", "-d", str(out), "--", sys.executable, str(script)],
                           env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        text = ""
        for dirpath, _, files in os.walk(out):
            for f in files:
                if "marker" in f and f.endswith(".csv"):
                    text += open(os.path.join(dirpath, f), errors="replace").read()
        assert "pdhg_trial_step" in text and "pdhg:A*xbar + dual step (K3+K4)" in text, text[:800]
