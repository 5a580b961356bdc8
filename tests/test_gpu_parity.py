"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bar (BASELINE.json north_star): status vectors bit-exact, iteration counts equal,
z within 1e-10 relative (Float64)."""
import numpy as np
import pytest

from conftest import assert_parity, colmajor, oracle_batch

pytestmark = pytest.mark.gpu


def run_cfg(pkg, orc, cfg, nprob, seed0=None, rtol=1e-10):
    prob = pkg.generate_batch(cfg, nprob, seed0 if seed0 is not None else pkg.BASE_SEED)
    x0, S0, st = pkg.phase1_batch(prob)
    assert (st == 1).all()
    z, S, status, detail, stats = pkg.solveQP_batch(prob, S0, x0, want_stats=True)
    zo, So, sto, deto, _ = oracle_batch(orc, prob, S0, x0)
    assert (sto > 0).all()
    rel = assert_parity(z, S, status, zo, So, sto, rtol)
    assert (detail == 0).all()
    assert np.array_equal(stats["iters"], sto)
    return rel, stats


def test_reference_kat_3x3(pkg):
    """test/runtests.jl:23-32 -- the only result the reference pins on this path."""
    V = np.array([[1 / 100, 1 / 80, 1 / 100], [1 / 80, 1 / 16, 1 / 40], [1 / 100, 1 / 40, 1 / 25]])
    Q = pkg.QP(V, u=np.array([0.7, np.inf, 0.7]))
    z, S, it = pkg.solveQP(Q)
    assert S.tolist() == [pkg.UP, pkg.IN, pkg.IN]
    assert it == 2
    np.testing.assert_allclose(z, [0.7, 11 / 210, 52 / 210], rtol=1e-12)


def test_cfg1_n50(pkg, orc):
    rel, stats = run_cfg(pkg, orc, pkg.CONFIGS["cfg1"], 16)
    assert (stats["path"] == 1).all()


def test_small_with_inequalities(pkg, orc):
    cfg = pkg.GenConfig(64, 1, 4, 128, 1e-3, 0.12, 1.05, 0.1)
    run_cfg(pkg, orc, cfg, 32)


def test_odd_n_scalar_loads(pkg, orc):
    cfg = pkg.GenConfig(51, 1, 3, 100, 1e-3, 0.2, 1.02, 0.1)
    run_cfg(pkg, orc, cfg, 8)


def test_multi_equalities(pkg, orc):
    cfg = pkg.GenConfig(96, 3, 5, 200, 1e-3, 0.1, 1.02, 0.1)
    run_cfg(pkg, orc, cfg, 8)


def test_cfg2_n512(pkg, orc):
    rel, stats = run_cfg(pkg, orc, pkg.CONFIGS["cfg2"], 8)
    assert (stats["path"] == 1).all()


def test_cfg3_n256_large_k_global_arena(pkg, orc):
    rel, stats = run_cfg(pkg, orc, pkg.CONFIGS["cfg3"], 2)
    assert stats["max_k"].max() > 180


def test_trace_matches_oracle(pkg, orc):
    cfg = pkg.GenConfig(64, 1, 4, 128, 1e-3, 0.12, 1.05, 0.1)
    prob = pkg.generate_batch(cfg, 4)
    x0, S0, st = pkg.phase1_batch(prob)
    db = pkg.DeviceBatch(prob, S0, x0, ntrace=512)
    db.solve()
    res = db.results()
    for p in range(4):
        A = colmajor(prob["A"][p], cfg.M)
        G = colmajor(prob["G"][p], cfg.J)
        zo, So, sto, det, tr = orc.solveQP_warm(prob["V"][p], A, G, prob["q"][p], prob["b"][p], prob["g"][p],
                                                prob["d"][p], prob["u"][p], S0[p], x0[p], max_trace=512)
        assert res["status"][p] == sto
        got = [tuple(int(v) for v in r) for r in res["trace"][p][:sto]]
        assert got == tr
