"""-m gpu: round-3 additions -- multipliers across the boundary, the timed configuration of bench.py against the
oracle on every problem, lazy hand-over through the host-buffer entry points, a Phase-1 LP that must return."""
import glob
import os

import numpy as np
import pytest

from conftest import assert_parity, colmajor, oracle_batch

pytestmark = pytest.mark.gpu

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def _mult_close(lam, gam, lamo, gamo, rtol=1e-10):
    """north_star: "x/lambda within 1e-10 rel" -- relative to the largest multiplier of the problem"""
    for got, ref in ((lam, lamo), (gam, gamo)):
        if ref.size == 0:
            continue
        scale = np.maximum(np.abs(ref).max(axis=-1, keepdims=True), 1e-3)   # (multipliers of a portfolio QP are ~1e-2)
        rel = (np.abs(got - ref) / scale).max()
        assert rel < rtol, rel


@pytest.mark.parametrize("name,nprob", [("cfg2", 8), ("cfg4", 96), ("cfg1", 32), ("cfg3", 12)])
def test_multipliers_match_oracle(pkg, orc, name, nprob):
    """alphaL (SSQP.jl:351) and gamma (:352) of the last pass leave the library by row / variable id and agree with the
    oracle's to 1e-10 relative -- both product kernels and both builds of the wavefront kernel"""
    cfg = pkg.CONFIGS[name]
    prob = pkg.generate_batch(cfg, nprob)
    x0, S0, st = pkg.phase1_batch(prob)
    zo, So, sto, _, _, lamo, gamo = orc.solveQP_warm_batch(prob["V"], prob["A"], prob["G"], prob["q"], prob["b"],
                                                          prob["g"], prob["d"], prob["u"], S0, x0, want_mult=True)
    assert (sto > 0).all()
    ctx = pkg.default_context()
    for opts in (dict(), dict(wave_qp_per_cu=8), dict(wave_kernel=0)):
        with ctx.options(**opts):
            z, S, status, detail, lam, gam = pkg.solveQP_batch(prob, S0, x0, want_mult=True)
        assert_parity(z, S, status, zo, So, sto)
        _mult_close(lam, gam, lamo, gamo)
    # the sign conventions of the reference at the optimum (KKTchk! found nothing to release, SSQP.jl:139-171)
    tolG = 2.0 ** -33
    N = cfg.N
    up, dn = So[:, :N] == 2, So[:, :N] == 1
    # (polishSz! may relabel an IN variable that sits on a bound: those have no multiplier -- gamma == 0 there)
    assert (gam[up] <= tolG).all() and (gam[dn] >= -tolG).all()
    if cfg.J:
        assert (lam[:, cfg.M:] >= -tolG).all()


def test_multipliers_with_purged_rows_and_goldens(pkg, orc):
    """duplicate inequality rows: the rank filter purges one of them and KKTchk! computes its multiplier by least
    squares (SSQP.jl:158-159); the golden fixtures cover equality blocks, free variables and the K == 0 exit"""
    rng = np.random.default_rng(20261004)
    cfg = pkg.GenConfig(64, 1, 6, 128, 1e-3, 0.07, 0.97, 0.1)
    prob = pkg.generate_batch(cfg, 24, 99)
    G = prob["G"].reshape(24, 64, 6)
    G[:, :, 1] = G[:, :, 0]
    prob["g"][:, 1] = prob["g"][:, 0]
    x0, S0, st = pkg.phase1_batch(prob)
    ok = st == 1
    sub = {k: np.ascontiguousarray(v[ok]) for k, v in prob.items()}
    zo, So, sto, _, _, lamo, gamo = orc.solveQP_warm_batch(sub["V"], sub["A"], sub["G"], sub["q"], sub["b"], sub["g"],
                                                          sub["d"], sub["u"], S0[ok], x0[ok], want_mult=True)
    conv = sto > 0
    assert conv.sum() >= 8
    ctx = pkg.default_context()
    for opts in (dict(), dict(wave_kernel=0)):
        with ctx.options(**opts):
            z, S, status, detail, lam, gam = pkg.solveQP_batch(sub, S0[ok], x0[ok], want_mult=True)
        assert np.array_equal(status, sto) and np.array_equal(S[conv], So[conv])
        _mult_close(lam[conv], gam[conv], lamo[conv], gamo[conv], rtol=1e-9)   # (least squares of a rank-deficient pair)
    for path in GOLDEN:
        d = np.load(path)
        c = {k: d[k] for k in d.files}
        if int(c["phase1_status"]) != 1 or int(c["status"]) <= 0:
            continue
        one = dict(V=c["V"][None], A=np.ascontiguousarray(c["A"].T)[None], G=np.ascontiguousarray(c["G"].T)[None],
                   q=c["q"][None], b=c["b"][None], g=c["g"][None], d=c["d"][None], u=c["u"][None])
        S0g, x0g = c["S0"][None].astype(np.int32), c["x0"][None]
        zo, So, sto, _, _, lamo, gamo = orc.solveQP_warm_batch(one["V"], one["A"], one["G"], one["q"], one["b"],
                                                              one["g"], one["d"], one["u"], S0g, x0g, want_mult=True)
        for opts in (dict(), dict(wave_kernel=0)):
            with ctx.options(**opts):
                z, S, status, detail, lam, gam = pkg.solveQP_batch(one, S0g, x0g, want_mult=True)
            assert np.array_equal(status, sto) and np.array_equal(S, So), path
            _mult_close(lam, gam, lamo, gamo, rtol=1e-9)


def test_library_multipliers_pass_a_tight_kkt_check(pkg):
    """the solver decides at tolG = 2^-33: with the library's OWN multipliers the KKT residuals of every cfg4 solution
    are checked at 1e-9 (stationarity of all N variables, not only the free ones) and the sign tests at tolG"""
    cfg = pkg.CONFIGS["cfg4"]
    prob = pkg.generate_batch(cfg, 128)
    x0, S0, st = pkg.phase1_batch(prob)
    z, S, status, detail, lam, gam = pkg.solveQP_batch(prob, S0, x0, want_mult=True)
    assert (status > 0).all()
    tolG = 2.0 ** -33
    N, M, J = cfg.N, cfg.M, cfg.J
    for p in range(128):
        V = prob["V"][p]
        C = np.vstack([colmajor(prob["A"][p], M), colmajor(prob["G"][p], J)])
        # stationarity: V z + q + [A;G]' lambda = gamma (gamma = 0 on the free variables)
        res = V @ z[p] + prob["q"][p] + C.T @ lam[p] - gam[p]
        assert np.abs(res).max() < 1e-9, (p, np.abs(res).max())
        Sz, Se = S[p][:N], S[p][N:]
        assert (gam[p][Sz == 2] <= tolG).all() and (gam[p][Sz == 1] >= -tolG).all()
        assert (lam[p][M:] >= -tolG).all()
        slack = prob["g"][p] - C[M:] @ z[p]
        assert (np.abs(lam[p][M:][slack > 1e-7]) == 0).all()        # complementarity: inactive rows carry no multiplier
        assert (slack > -1e-9).all() and abs(C[:M] @ z[p] - prob["b"][p]).max() < 1e-9


def test_timed_configuration_all_1024_against_oracle(pkg, orc):
    """exactly what bench.py times: cfg4 x 1024 per lane, three launch lanes with DISTINCT batches (own seeds), the
    eight-per-CU build (wave_qp_per_cu = 8), lazy hand-over, overlapping launches on three streams -- every one of
    the 3 x 1024 problems against the oracle"""
    import torch
    cfg = pkg.CONFIGS["cfg4"]
    nprob, nl = 1024, 3
    dev = torch.device("cuda", pkg.default_context().device)
    lanes = []
    for i in range(nl):
        c = pkg.Context(dev.index)
        c.set_option("lazy_handover", 1)
        c.set_option("wave_qp_per_cu", 8)
        b, prob, x0, S0 = pkg.DeviceBatch.generated(cfg, nprob, pkg.BASE_SEED + i * nprob, ctx=c, device=dev.index)
        lanes.append((b, torch.cuda.Stream(dev), prob, x0, S0))
    for rep in range(2):
        for b, st, *_ in lanes:
            with torch.cuda.stream(st):
                b.solve()
    for b, st, *_ in lanes:
        b.ctx.sync(st.cuda_stream)
    torch.cuda.synchronize(dev)
    for b, st, prob, x0, S0 in lanes:
        r = b.results()
        full = dict(prob)
        full["V"] = b.t["V"].cpu().numpy()
        zo, So, sto, _, _ = oracle_batch(orc, full, S0, x0)
        assert_parity(r["z"], r["S"], r["status"], zo, So, sto)
        assert ((r["stats"]["path"] & 16) != 0).all()            # the wavefront kernel ran every QP
    assert not np.array_equal(lanes[0][0].results()["S"], lanes[1][0].results()["S"])   # really different batches


def test_lazy_handover_host_buffer_entries(pkg, orc):
    """lazy_handover = 1 on the CALLER's context through the host-buffer entry points (one chunk, so the call runs on
    that context itself) and through the resident handle: cfg3 hands every QP over, and the owed launch has to go
    out before the results are copied back"""
    cfg = pkg.CONFIGS["cfg3"]
    prob = pkg.generate_batch(cfg, 40, 31337)
    x0, S0, st = pkg.phase1_batch(prob)
    zo, So, sto, _, _ = oracle_batch(orc, prob, S0, x0)
    ctx = pkg.Context(pkg.default_context().device)
    ctx.set_option("lazy_handover", 1)
    z, S, status, detail, stats = pkg.solveQP_batch(prob, S0, x0, ctx=ctx, want_stats=True)
    assert_parity(z, S, status, zo, So, sto)
    rb = pkg.ResidentBatch(prob, ctx=ctx)
    z2, S2, st2, _ = rb.solve(S0, x0)
    assert_parity(z2, S2, st2, zo, So, sto)
    z3, S3, st3, _ = rb.solve(S0, x0)          # again: nothing stale is left from the call before
    assert np.array_equal(z3, z2) and np.array_equal(S3, S2) and np.array_equal(st3, st2)
    rb.close()


def test_phase1_gpu_returns_on_poisoned_lp(pkg):
    """a NaN in A / G: every comparison that involves it is false, so the poisoned column never becomes a candidate
    (its reduced cost is NaN) and never enters the xb sum (it sits at a zero bound): the host stage solves the LP around
    it.  The GPU stage must come back -- not spin -- with exactly the host stage's status, x0 and S0."""
    import torch
    cfg = pkg.GenConfig(40, 1, 3, 80, 1e-3, 0.1, 1.0, 0.1)
    prob = pkg.generate_batch(cfg, 4, 5)
    prob["G"][1, 3, 0] = np.nan
    prob["A"][2, 5, 0] = np.nan
    P, N = prob["q"].shape
    xh, Sh, sth = pkg.phase1_batch(prob)
    db = pkg.DeviceBatch(prob, np.zeros((P, N + cfg.J), dtype=np.int32), np.zeros((P, N)))
    st = db.phase1()
    torch.cuda.synchronize()
    st = st.cpu().numpy()
    assert st[0] == 1 and st[3] == 1           # the clean problems are untouched
    assert np.array_equal(st, sth), (st, sth)
    assert np.array_equal(db.S0.cpu().numpy(), Sh)
    assert np.array_equal(db.x0.cpu().numpy(), xh, equal_nan=True)


# ---------------------------------------------------------------- the big-factor build of the wavefront kernel
def _oracle_fast(orc, prob, S0, x0):
    return orc.solveQP_warm_batch(prob["V"], prob["A"], prob["G"], prob["q"], prob["b"], prob["g"], prob["d"],
                                  prob["u"], S0, x0, lapack=orc.lapack_available())


@pytest.mark.parametrize("family", ["cfg3_like", "blocking", "inequalities", "n512"])
def test_big_factor_build_families(pkg, orc, family):
    """free sets of 128..250 rows in the four-row-slot build (rows >= 64 in global scratch by columns and by rows, every
    stream through the LDS ring): appends only (cfg3), upper bounds that block at large K (streamed deletes across the
    slot boundaries), active inequalities at large K, N = 512.  Reached by hand-over from both first-stage builds and from the first pass."""
    if family == "cfg3_like":       # K -> ~170, appends almost only
        cfg, n = pkg.GenConfig(200, 1, 0, 400, 1e-3, 0.1, 1.2, 0.0), 24
    elif family == "blocking":      # K -> ~215 with ~60 blocked steps on the way (deletes in every row slot)
        cfg, n = pkg.GenConfig(256, 1, 0, 512, 1e-3, 2.5 / 256, 1.2, 0.0), 24
    elif family == "inequalities":  # K -> ~195 with up to 10 active rows and ~40 blocked steps
        cfg, n = pkg.GenConfig(224, 2, 8, 448, 1e-3, 4.0 / 224, 1.0, 0.0), 24
    else:                           # N = 512: K -> ~180, ~90 blocked steps, up to 5 active rows
        cfg, n = pkg.GenConfig(512, 1, 4, 1024, 1e-3, 8.0 / 512, 1.0, 0.01), 12
    prob = pkg.generate_batch(cfg, n, 424242)
    x0, S0, st = pkg.phase1_batch(prob)
    ok = st == 1
    assert ok.sum() >= n // 2
    sub = {k: np.ascontiguousarray(v[ok]) for k, v in prob.items()}
    zo, So, sto, _, _ = _oracle_fast(orc, sub, S0[ok], x0[ok])
    assert (sto > 0).all()
    ctx = pkg.default_context()
    seen_big = False
    for opts in (dict(), dict(wave_qp_per_cu=8), dict(wave_kernel=2)):
        with ctx.options(**opts):
            z, S, status, detail, stats = pkg.solveQP_batch(sub, S0[ok], x0[ok], want_stats=True)
        assert_parity(z, S, status, zo, So, sto)
        assert (stats["max_k"] > 127).any(), stats["max_k"]
        big = (stats["max_k"] > 127) & (stats["max_k"] <= 252)
        assert ((stats["path"][big] & 32) == 0).all()        # the workgroup kernel was not needed for those
        seen_big = seen_big or bool(big.any())
    assert seen_big


def test_big_factor_build_trace_and_multipliers(pkg, orc):
    """per-pass trace (K, W, kind, id) and the multipliers of the last pass through the hand-over chain"""
    cfg = pkg.GenConfig(192, 1, 3, 384, 1e-3, 3.0 / 192, 1.0, 0.0)    # K -> ~160, up to 4 active rows, ~45 blocked steps
    prob = pkg.generate_batch(cfg, 6, 424242)
    x0, S0, st = pkg.phase1_batch(prob)
    assert (st == 1).all()
    zo, So, sto, _, _, lamo, gamo = orc.solveQP_warm_batch(prob["V"], prob["A"], prob["G"], prob["q"], prob["b"],
                                                          prob["g"], prob["d"], prob["u"], S0, x0, want_mult=True)
    db = pkg.DeviceBatch(prob, S0, x0, ntrace=1024).want_multipliers()
    db.solve()
    r = db.results()
    assert_parity(r["z"], r["S"], r["status"], zo, So, sto)
    assert (r["stats"]["max_k"] > 127).any()
    _mult_close(r["lam"], r["gam"], lamo, gamo)
    for p in range(6):
        A = colmajor(prob["A"][p], cfg.M)
        G = colmajor(prob["G"][p], cfg.J)
        _, _, sto1, _, tr = orc.solveQP_warm(prob["V"][p], A, G, prob["q"][p], prob["b"][p], prob["g"][p],
                                             prob["d"][p], prob["u"][p], S0[p], x0[p], max_trace=1024)
        got = [tuple(int(v) for v in row) for row in r["trace"][p][:sto1]]
        assert got == tr
