"""The linear-approximation fallback and the accept rule: oracle vs the fixture captured from the
reference's LinearApproximationSolver (oracle/capture_golden_fallback.py)."""
import os

import numpy as np

from oracle import checks_np as CK
from oracle import fallback_np as FB

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cases():
    g = np.load(os.path.join(GOLD, "fallback_linear.npz"))
    for k in range(len(g["n"])):
        n, m = int(g["n"][k]), int(g["m"][k])
        yield k, n, m, g


def test_linear_approximation_matches_the_reference_bit_for_bit():
    for k, n, m, g in cases():
        out = FB.linear_approximation(g["is_slack"][k, :n], g["loads"][k, :n], g["gens"][k, :n], g["total_load"][k], g["total_gen"][k],
                                      g["line_from"][k, :m], g["line_to"][k, :m], g["line_x"][k, :m], g["line_rating"][k, :m])
        assert np.array_equal(out["bus_voltages"], g["bus_voltages"][k, :n]), k
        assert np.array_equal(out["bus_angles"], g["bus_angles"][k, :n]), k
        assert np.array_equal(out["line_flows"], g["line_flows"][k, :m]), k
        assert np.array_equal(out["line_loadings"], g["line_loadings"][k, :m]), k
        assert out["losses"] == g["losses"][k], k


def test_quality_of_the_fallback_answer_and_the_accept_rule():
    qs = []
    for k, n, m, g in cases():
        q = CK.quality(np.array([True]), np.array([1]), np.array([0.0]), g["bus_voltages"][k:k + 1, :n], g["line_loadings"][k:k + 1, :m],
                       g["line_flows"][k:k + 1, :m], 1e-6)
        assert q[0] == g["quality"][k], k
        qs.append(q[0])
    qs = np.array(qs)
    primary = np.where(np.arange(len(qs)) % 2 == 0, 0.0, 1.0)          # every other primary answer was rejected
    method = FB.accept_or_fall_back(primary, qs)
    assert ((method == 0) == (primary > 0.7)).all()
    assert ((method == 1) == ((primary <= 0.7) & (qs > 0.7))).all()
    assert (method == -1).any() and (method == 1).any()
