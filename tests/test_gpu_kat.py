"""The same reference KATs through the HIP engine (C ABI) on the GPU."""
import pytest

from firstorderlp_jl_amd import HipPdhgEngine
from tests import kat_common

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", kat_common.CASES, ids=lambda c: c.__name__)
def test_reference_kat_on_hip(gpu_required, case):
    case(HipPdhgEngine.from_problem)
