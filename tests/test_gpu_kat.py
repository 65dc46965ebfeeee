"""The same reference KATs (test/test_primal_dual_hybrid_gradient.jl:77-423)
through the HIP engine (C ABI) on the GPU -- once with the layout the library
picks for these tiny LPs (stream), and once each with the TILED layout forced
at two tile widths, so that ``spmv_tiled_kernel`` -- the kernel bench.py
times -- meets the reference-held values directly."""
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from tests import kat_common

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", kat_common.CASES, ids=lambda c: c.__name__)
def test_reference_kat_on_hip(gpu_required, case):
    case(HipPdhgEngine.from_problem)


def _tiled_factory(problem, **kw):
    eng = HipPdhgEngine.from_problem(problem, **kw)
    info = eng.layout_info()
    assert info["A_tiled_waves"] > 0 and info["At_tiled_waves"] > 0, info
    return eng


@pytest.mark.parametrize("shift", [8, 10])
@pytest.mark.parametrize("case", kat_common.CASES, ids=lambda c: c.__name__)
def test_reference_kat_on_hip_tiled_layout(gpu_required, monkeypatch, case, shift):
    monkeypatch.setenv("PDHG_SPMV", "tiled")
    monkeypatch.setenv("PDHG_TILE_SHIFT", str(shift))
    case(_tiled_factory)
