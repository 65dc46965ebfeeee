"""tools/selfprof.py (bench.py's self-measured roofline evidence): the arithmetic that turns rocprofv3's databases into
`kernel_ms_rocprof` and `traffic`, on synthetic databases with the two views it reads (no GPU, no rocprofv3)."""
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tools import selfprof  # noqa: E402

FULL = "void (anonymous namespace)::%s((anonymous namespace)::CsrView, double const*)"


def _kt_db(path, rows):
    con = sqlite3.connect(path)
    con.execute("create table top_kernels (name text, total_calls int, total_duration real, average real, percentage real)")
    con.executemany("insert into top_kernels values (?, ?, ?, ?, ?)", rows)
    con.commit()
    con.close()


def _pmc_db(path, rows):
    con = sqlite3.connect(path)
    con.execute("create table counters_collection (kernel_name text, counter_name text, value real)")
    con.executemany("insert into counters_collection values (?, ?, ?)", rows)
    con.commit()
    con.close()


def test_product_time_adds_the_kernels_of_a_fused_product(tmp_path):
    d = tmp_path / "kt" / "host" / "1"
    d.mkdir(parents=True)
    _kt_db(str(d / "x_results.db"), [
        (FULL % "spmv_stream_kernel<0, false, 1>", 6, 300.0, 50.0, 40.0),
        (FULL % "spmv_stream_kernel<2, true, 1>", 6, 204.0, 34.0, 30.0),
        (FULL % "spmv_long_partial_kernel<1>", 6, 54.0, 9.0, 5.0),
        (FULL % "spmv_long_final_kernel<2>", 6, 30.0, 5.0, 3.0),
        (FULL % "primal_kernel<false, true>", 6, 72.0, 12.0, 10.0)])
    times = selfprof.kernel_times(str(tmp_path / "kt"))
    assert times["spmv_stream_kernel<0, false, 1>"] == (6, 50.0)
    product = "spmv_stream_kernel<0, false, 1> + spmv_stream_kernel<2, true, 1> + spmv_long_partial_kernel<1> + spmv_long_final_kernel<2>"
    assert selfprof.product_time_us(times, product) == 50.0 + 34.0 + 9.0 + 5.0
    assert selfprof.product_time_us(times, "spmv_tiled_kernel<1, 0>") is None          # not in the trace


def test_traffic_counts_reads_by_request_size_and_writes(tmp_path):
    for i, rows in enumerate([
            [(FULL % "spmv_tiled_kernel<1, 0>", "TCC_EA0_RDREQ_sum", 1000.0), (FULL % "spmv_tiled_kernel<1, 0>", "TCC_EA0_RDREQ_sum", 1000.0),
             (FULL % "spmv_tiled_kernel<1, 0>", "TCC_EA0_RDREQ_32B_sum", 100.0), (FULL % "spmv_tiled_kernel<1, 0>", "TCC_EA0_RDREQ_32B_sum", 100.0),
             (FULL % "spmv_tiled_kernel<1, 0>", "TCC_EA0_RDREQ_128B_sum", 600.0), (FULL % "spmv_tiled_kernel<1, 0>", "TCC_EA0_RDREQ_128B_sum", 600.0),
             (FULL % "spmv_tiled_kernel<1, 0>", "TCC_HIT_sum", 5.0), (FULL % "spmv_tiled_kernel<1, 0>", "TCC_HIT_sum", 7.0)],
            [(FULL % "spmv_tiled_kernel<1, 0>", "WRITE_SIZE", 2.0), (FULL % "spmv_tiled_kernel<1, 0>", "WRITE_SIZE", 2.0),
             (FULL % "spmv_tiled_kernel<1, 0>", "TCC_MISS_sum", 11.0), (FULL % "spmv_tiled_kernel<1, 0>", "TCC_MISS_sum", 11.0)]]):
        d = tmp_path / f"p{i + 1}"
        d.mkdir()
        _pmc_db(str(d / "r.db"), rows)
    ctr = {}
    for i in (1, 2):
        for k, v in selfprof.counters(str(tmp_path / f"p{i}")).items():
            ctr.setdefault(k, {}).update(v)
    traffic, detail = selfprof.product_traffic_bytes(ctr, "spmv_tiled_kernel<1, 0>")
    # reads = 32 * 100 + 128 * 600 + 64 * (1000 - 700); writes = 1024 * 2  (per launch: the averages over the two dispatches)
    assert traffic == 32 * 100 + 128 * 600 + 64 * 300 + 2048
    assert detail["spmv_tiled_kernel<1, 0>"]["launches"] == 2 and detail["spmv_tiled_kernel<1, 0>"]["l2_hits"] == 6
    assert selfprof.product_traffic_bytes(ctr, "spmv_tiled_kernel<2, 0>") == (None, None)


def test_run_without_rocprofv3_reports_it(monkeypatch):
    monkeypatch.setattr(selfprof.shutil, "which", lambda name: None)
    assert selfprof.run("random", "spmv_tiled_kernel<1, 0>") == {"error": "rocprofv3 not on PATH"}
