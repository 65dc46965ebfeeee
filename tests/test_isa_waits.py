"""The ISA of the product kernels must keep its memory requests INDEPENDENT where the source says so (round 6, NOTEBOOK.md 10.1-10.2).

Two compiler accidents cost 12-17 % for rounds without showing in any test: a sign extension hoisted behind every index load
of the CSR stream kernel put `s_waitcnt vmcnt(1)` behind each (col, val) pair -- eight dependent round trips to HBM per row
block since round 1 -- and the restructured sliced jagged loop got a wait in front of every gather (the wait-count pass cannot
count conditionally issued loads).  This test compiles the device code (hipcc cross-compiles without a GPU, ~30 s) and checks
the shape of the instruction stream: somewhere in spmv_stream_kernel sixteen non-temporal entry loads go out with no vmcnt wait
between them, and the sliced jagged kernels issue a batch's sixteen gathers back to back."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    hipcc = shutil.which(os.environ.get("HIPCC", "hipcc"))
    if not hipcc:
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "pdhg.s"
    r = subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                        "--cuda-device-only", "-S", "-o", str(out), os.path.join(ROOT, "firstorderlp.jl_amd", "csrc", "pdhg_hip.hip")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


def _body(isa, key):
    m = re.search(r"\n(_ZN[^\n:]*" + re.escape(key) + r"[^\n:]*):[^\n]*\n(.*?)\n\s*s_endpgm", isa, re.S)
    assert m, key
    return m.group(2)


def _longest_run(body, is_item):
    """Longest run of matching loads with no `s_waitcnt vmcnt` (and no barrier) between them."""
    best = run = 0
    for line in body.split("\n"):
        t = line.strip()
        if is_item(t):
            run += 1
            best = max(best, run)
        elif t.startswith("s_waitcnt") and "vmcnt" in t or t.startswith("s_barrier"):
            run = 0
    return best


def test_stream_kernel_issues_its_sixteen_entry_loads_together(isa):
    for key in ("spmv_stream_kernelILi1ELb0ELi0E", "spmv_stream_kernelILi2ELb0ELi1E", "spmv_stream_kernelILi0ELb0ELi0E"):
        run = _longest_run(_body(isa, key), lambda t: t.startswith("global_load_dword") and t.endswith(" nt"))
        assert run >= 16, f"{key}: the (col, val) loads of a row block are issued {run} at a time (16 expected: unsigned column offsets)"


def test_sliced_jagged_kernels_issue_a_batch_of_gathers_back_to_back(isa):
    for key in ("spmv_sj_kernelILi1ELb0ELi0ELi1E", "spmv_sj_kernelILi2ELb0ELi1ELi8E", "spmv_sj_kernelILi2ELb0ELi1ELi1E", "spmv_sj_kernelILi1ELb0ELi0ELi8E"):
        run = _longest_run(_body(isa, key), lambda t: t.startswith("global_load_dwordx2") and not t.endswith(" nt"))
        assert run >= 12, f"{key}: {run} gathers in flight (the explicit wait at the top of a batch is gone or no longer understood)"


def test_pipelined_stream_kernel_gathers_back_to_back(isa):
    run = _longest_run(_body(isa, "spmv_stream_pipe_kernelILi1ELb0ELi0E"), lambda t: t.startswith("global_load_dwordx2") and not t.endswith(" nt"))
    assert run >= 8
