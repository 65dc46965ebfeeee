import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import folp_loader  # noqa: E402

folp_loader.load()


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The bitwise comparisons with the CPU oracle hold for EVERY row only in strict row order
# (PDHG_ROW_ORDER=strict: each row added left to right by one lane).  The library's default is
# "relaxed" (rows of more than 64 entries are summed wave-parallel, within 1e-13 * sum |a x|): the
# suite runs strict unless a test asks for relaxed itself (tests/test_gpu_row_order.py does, and
# __graft_entry__.smoke() / bench.py run the default).
os.environ.setdefault("PDHG_ROW_ORDER", "strict")


def pytest_sessionstart(session):
    """Build what the tests load if it is missing or stale (the same steps as
    __graft_entry__.build(); hipcc cross-compiles without a GPU, both are no-ops
    when up to date)."""
    import shutil
    from firstorderlp_jl_amd import _lib
    from oracle import oracle
    # The HIP library is only needed by the gpu-marked tests (and the symbol check):
    # without hipcc on this box leave whatever library is there and let THOSE tests
    # fail loudly; the pure-host tests must still run.
    if shutil.which(os.environ.get("HIPCC", "hipcc")):
        _lib.build()
    oracle.build()
    oracle.build_omp()


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_required():
    """GPU tests must fail loudly (not skip) when selected without a GPU."""
    if not has_gpu():
        pytest.fail("test marked gpu but no GPU is visible")


def pytest_collection_modifyitems(config, items):
    """Every GPU test gets a timeout (10 minutes unless it sets its own): a test that
    hangs -- or a helper that quietly asks for a 150 GiB permutation -- must not burn
    the GPU box's time."""
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(600))
