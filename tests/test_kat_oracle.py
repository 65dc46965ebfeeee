"""Pins the CPU oracle (and the host driver) against every PDHG known-answer
test of the reference: test/test_primal_dual_hybrid_gradient.jl.  CPU only."""
import pytest

from tests import kat_common
from tests.oracle_engine import OracleEngine


@pytest.mark.parametrize("case", kat_common.CASES, ids=lambda c: c.__name__)
def test_reference_kat_on_oracle(case):
    case(OracleEngine.from_problem)
