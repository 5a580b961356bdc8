"""-m "not gpu": the oracle against the reference's known answer, the golden fixtures, the numpy/LAPACK
restatement and the KKT verifier.  No GPU needed."""
import glob
import os

import numpy as np
import pytest

from conftest import colmajor
from kkt import assert_kkt

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
IN, DN, UP, OE, EO = 0, 1, 2, 3, 4


def load(path):
    d = np.load(path)
    return {k: d[k] for k in d.files}


def test_reference_known_answer(orc):
    """test/runtests.jl:23-32: solveQP(QP(V; u=[0.7,Inf,0.7])) gives S == [UP, IN, IN]; the hand trace of the
    source adds x0=[0.7,0.3,0], S0=[UP,IN,DN], 2 iterations, z=[0.7,11/210,52/210]."""
    V = np.array([[1 / 100, 1 / 80, 1 / 100], [1 / 80, 1 / 16, 1 / 40], [1 / 100, 1 / 40, 1 / 25]])
    A, b, G, g = np.ones((1, 3)), [1.0], np.zeros((0, 3)), []
    d, u, q = np.zeros(3), [0.7, np.inf, 0.7], np.zeros(3)
    x0, S0, st = orc.initQP(A, G, b, g, d, u)
    assert st == 1 and S0.tolist() == [UP, IN, DN]
    np.testing.assert_allclose(x0, [0.7, 0.3, 0.0], atol=1e-15)
    z, S, status, det, tr = orc.solveQP(V, A, G, q, b, g, d, u, max_trace=8)
    assert S.tolist() == [UP, IN, IN]
    assert status == 2 and det == 0
    assert tr == [(1, 1, 2, 3), (2, 1, 3, 0)]      # iteration 1 releases variable 3, iteration 2 is optimal
    np.testing.assert_allclose(z, [0.7, 11 / 210, 52 / 210], rtol=1e-13)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(orc, path):
    c = load(path)
    x0, S0, st = orc.initQP(c["A"], c["G"], c["b"], c["g"], c["d"], c["u"])
    assert st == int(c["phase1_status"])
    assert np.array_equal(S0, c["S0"]) and np.array_equal(x0, c["x0"])
    if st != 1:
        z, S, status, det, _ = orc.solveQP(c["V"], c["A"], c["G"], c["q"], c["b"], c["g"], c["d"], c["u"])
        assert status == 0                                     # infeasible: SSQP.jl:533-537
        return
    z, S, status, det, tr = orc.solveQP_warm(c["V"], c["A"], c["G"], c["q"], c["b"], c["g"], c["d"], c["u"],
                                            c["S0"], c["x0"], max_trace=4096)
    assert status == int(c["status"]) and det == 0
    assert np.array_equal(S, c["S"])
    assert np.array_equal(np.array(tr, dtype=np.int32).reshape(-1, 4), c["trace"])
    np.testing.assert_allclose(z, c["z"], rtol=0, atol=1e-13 * max(1.0, np.abs(c["z"]).max()))
    assert_kkt(c["V"], c["A"], c["G"], c["q"], c["b"], c["g"], c["d"], c["u"], z, S)


def test_oracle_matches_lapack_restatement(pkg, orc):
    """plain-C oracle vs the numpy restatement that calls potrf/potri like Julia's LinearAlgebra."""
    from oracle import ssqp_numpy as onp
    for N, M, J, ub, gs, qs, seed in [(24, 1, 0, 0.0, 1.2, 0.0, 1), (40, 1, 4, 0.1, 1.0, 0.1, 2),
                                      (33, 2, 3, 0.12, 0.98, 0.2, 3), (64, 1, 6, 0.06, 0.95, 0.1, 4)]:
        cfg = pkg.GenConfig(N, M, J, 2 * N, 1e-3, ub, gs, qs)
        prob = pkg.generate_batch(cfg, 3, 777 + seed)
        x0, S0, st = pkg.phase1_batch(prob)
        for p in range(3):
            if st[p] != 1:
                continue
            A, G = colmajor(prob["A"][p], M), colmajor(prob["G"][p], J)
            a = orc.solveQP_warm(prob["V"][p], A, G, prob["q"][p], prob["b"][p], prob["g"][p], prob["d"][p],
                                 prob["u"][p], S0[p], x0[p], max_trace=4096)
            b = onp.solveQP_warm(prob["V"][p], A, G, prob["q"][p], prob["b"][p], prob["g"][p], prob["d"][p],
                                 prob["u"][p], S0[p], x0[p])
            assert a[2] == b[2] and a[2] > 0
            assert np.array_equal(a[1], b[1])
            assert a[4] == [tuple(int(v) for v in t) for t in b[3]]
            assert np.abs(a[0] - b[0]).max() < 1e-12


def test_getrows_gjr_edge_cases(orc):
    """src/utils.jl:49-86: dependent rows are dropped, the rhs column competes for pivots, absolute threshold."""
    from oracle import ssqp_numpy as onp
    tol = 2.0 ** -26
    X = np.array([[1.0, 1, 1, 1], [2, 2, 2, 2], [0, 1, 0, 3]])              # row 1 = 2 x row 0
    assert orc.getRowsGJr(X, tol)[0] == [0, 2] == onp.getRowsGJr(X, tol)[0]
    X = np.array([[1.0, 1, 1, 1], [2, 2, 2, 5]])                             # inconsistent: kept through the rhs
    assert orc.getRowsGJr(X, tol)[0] == [0, 1] == onp.getRowsGJr(X, tol)[0]
    X = np.array([[0.0, 0, 0, 0], [1e-9, 0, 0, 0], [0, 3, 0, 1]])            # zero row and a sub-threshold row
    assert orc.getRowsGJr(X, tol)[0] == [2] == onp.getRowsGJr(X, tol)[0]
    rng = np.random.default_rng(5)
    for _ in range(20):
        nr, nc = rng.integers(1, 7), rng.integers(2, 12)
        X = rng.standard_normal((nr, nc))
        if nr > 2:
            X[-1] = X[0] - 2 * X[1]
        assert orc.getRowsGJr(X, tol) == onp.getRowsGJr(X, tol)


def test_status_codes(orc):
    """return codes of solveQP (SSQP.jl:205-209, 226-228, 272-273)."""
    c = load([p for p in GOLDEN if p.endswith("box20_j3.npz")][0])
    args = (c["V"], c["A"], c["G"], c["q"], c["b"], c["g"], c["d"], c["u"])
    z, S, status, det, _ = orc.solveQP(*args, mc=-70)
    assert status == -1 and (S[:20] == DN).all() and (z == 0).all()
    st = orc.Settings(maxIter=5)
    z, S, status, det, _ = orc.solveQP_warm(*args, c["S0"], c["x0"], settings=st)
    assert status == -6                                        # -(maxIter+1): iter > maxIter -> -iter
    Vbad = c["V"] - 10.0 * np.eye(20)                          # indefinite: cholesky(V[F,F]) would throw
    z, S, status, det, _ = orc.solveQP_warm(Vbad, *args[1:], c["S0"], c["x0"])
    assert status == -1 and det == 1


def test_oracle_multipliers_satisfy_stationarity(pkg, orc):
    """the multipliers the oracle exports (alphaL SSQP.jl:351, gamma :352, purged rows :158-159) are checked by the
    mathematics, not by reading the code: V z + q + [A;G]' lambda = gamma on every variable, gamma = 0 on the free
    ones, signs as KKTchk! demands at an optimum, no multiplier on an inactive row"""
    from conftest import colmajor
    tolG = 2.0 ** -33
    for cfg, seed in [(pkg.GenConfig(40, 1, 0, 80, 1e-3, 3 / 32, 1.2, 0.0), 11),
                      (pkg.GenConfig(64, 1, 6, 128, 1e-3, 0.07, 0.97, 0.1), 12),
                      (pkg.GenConfig(48, 3, 5, 96, 1e-3, 0.1, 1.0, 0.1), 13)]:
        prob = pkg.generate_batch(cfg, 6, 3000 + seed)
        x0, S0, st = pkg.phase1_batch(prob, nthreads=2)
        z, S, status, _, _, lam, gam = orc.solveQP_warm_batch(prob["V"], prob["A"], prob["G"], prob["q"], prob["b"],
                                                             prob["g"], prob["d"], prob["u"], S0, x0, want_mult=True)
        assert (status > 0).all()
        N, M, J = cfg.N, cfg.M, cfg.J
        for p in range(6):
            C = np.vstack([colmajor(prob["A"][p], M), colmajor(prob["G"][p], J)])
            res = prob["V"][p] @ z[p] + prob["q"][p] + C.T @ lam[p] - gam[p]
            assert np.abs(res).max() < 1e-10, np.abs(res).max()
            Sz = S[p][:N]
            assert (gam[p][Sz == 0] == 0).all()
            assert (gam[p][Sz == 2] <= tolG).all() and (gam[p][Sz == 1] >= -tolG).all()
            assert (lam[p][M:] >= -tolG).all()
            slack = prob["g"][p] - C[M:] @ z[p]
            assert (lam[p][M:][slack > 1e-7] == 0).all()
