"""The independent KKT verifier itself (tests/kkt.py), on hand-made cases: it must accept an optimum at a degenerate
vertex (multipliers not unique: the least-norm ones are not the sign-feasible ones) and reject a non-optimal one."""
import numpy as np
import pytest

from kkt import assert_kkt, kkt_report, DN, UP


def _qp():
    V = np.eye(2)
    A = np.array([[1.0, 1.0]])
    G = np.zeros((0, 2))
    q = np.array([-3.0, -1.0])
    return V, A, G, q, np.array([1.0]), np.zeros(0), np.zeros(2), np.ones(2)


def test_degenerate_vertex_is_accepted():
    # min 1/2 |z|^2 - 3 z1 - z2, z1 + z2 = 1, 0 <= z <= 1: with z2 = 1 - z1 the objective is z1^2 - 3 z1 - 1/2,
    # decreasing on [0, 1]: the optimum is the vertex (1, 0) with NO free variable -- any multiplier in [1, 2] works,
    # the least-norm one (0) does not
    V, A, G, q, b, g, d, u = _qp()
    z, S = np.array([1.0, 0.0]), np.array([UP, DN])
    r = assert_kkt(V, A, G, q, b, g, d, u, z, S)
    assert r.get("multipliers", "").startswith("linear feasibility")


def test_non_optimal_vertex_is_rejected():
    V, A, G, q, b, g, d, u = _qp()
    z, S = np.array([0.0, 1.0]), np.array([DN, UP])     # the other vertex: feasible, not optimal
    r = kkt_report(V, A, G, q, b, g, d, u, z, S)
    assert r["gamma_dn_min"] < -0.5 or r["gamma_up_max"] > 0.5
    with pytest.raises(AssertionError):
        assert_kkt(V, A, G, q, b, g, d, u, z, S)
