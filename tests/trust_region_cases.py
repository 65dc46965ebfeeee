"""bound_optimal_objective known answers from test/test_trust_region_utils.jl:212-327,
shared by the host (numpy) and device (C ABI) implementations.
Each case: (problem maker, x, y, radius, norm, expected dict)."""
import numpy as np

from firstorderlp_jl_amd.trust_region_utils import EUCLIDEAN_NORM, MAX_NORM
from tests import helpers as H

CASES = [
    # :217-233  optimum: both bounds equal the optimal value
    ("opt_max", H.example_lp, [1.0, 0.0, 6.0, 2.0], [0.5, 4.0, 0.0], 10.0, MAX_NORM,
     dict(lower=-1.0, upper=-1.0)),
    # :235-249  slightly off the optimum
    ("near_opt", H.example_lp, [1.0, 0.0, 5.99999, 2.0], [0.50001, 4.0, 0.0], 10.0, MAX_NORM,
     dict(lower_in=(-1.01, -1.0), upper_in=(-1.0, -0.99))),
    # :252-268
    ("r2_max", H.example_lp, [1.0, 0.0, 6.0, 1.0], [0.0, 4.0, 0.0], 2.0, MAX_NORM,
     dict(lower=-4.0, upper=2.0)),
    # :270-291  joint ball
    ("euclid", H.example_lp, [3.0, 0.0, 6.0, 0.0], [0.0, 4.0, 0.0], 5.0, EUCLIDEAN_NORM,
     dict(lower=-4.0, lagrangian=-1.0, upper=7.0)),
    # :294-310  lower bound == corrected dual objective
    ("corrected", H.example_lp, [1.0, 1.0, 4.0, 1.0], [0.0, 0.0, 0.0], 10.0, MAX_NORM,
     dict(lower=-14.0)),
    # :312-326  interior point of the star LP: upper bound == Lagrangian
    ("cc_star", H.example_cc_star_lp, [0.5, 0.5, 0.5, 1.0, 1.0, 1.0], [0.0, 0.0, 0.0], 10.0, MAX_NORM,
     dict(upper_eq_lagrangian=True)),
]


def check(result, expected, tol):
    if "lower" in expected:
        assert abs(result.lower_bound_value - expected["lower"]) <= tol
    if "upper" in expected:
        assert abs(result.upper_bound_value - expected["upper"]) <= tol
    if "lagrangian" in expected:
        assert abs(result.lagrangian_value - expected["lagrangian"]) <= tol
    if "lower_in" in expected:
        assert expected["lower_in"][0] < result.lower_bound_value < expected["lower_in"][1]
        assert expected["upper_in"][0] < result.upper_bound_value < expected["upper_in"][1]
    if expected.get("upper_eq_lagrangian"):
        assert abs(result.lagrangian_value - result.upper_bound_value) <= tol
        assert result.lower_bound_value < result.lagrangian_value
