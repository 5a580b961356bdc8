"""-m gpu: round-4 additions -- the one-wavefront-per-QP Phase-1 and the single-launch solveQP(Q), stream switches with an
owed hand-over, multipliers of a QP that does not converge, the one-process multi-GPU entry over every device present."""
import numpy as np
import pytest

from conftest import assert_parity, oracle_batch

pytestmark = pytest.mark.gpu


def test_stream_switch_with_an_owed_handover(pkg, orc):
    """lazy_handover = 1 on cfg3 (every QP is handed over, so every call owes its later stages), consecutive solves of the
    SAME batch on DIFFERENT streams: the second call's reset of S must queue behind the owed launches of the first,
    which went out on the first call's stream (ssqp_flush_to)"""
    import torch
    cfg = pkg.CONFIGS["cfg3"]
    prob = pkg.generate_batch(cfg, 64, 777)
    x0, S0, st = pkg.phase1_batch(prob)
    zo, So, sto, _, _ = oracle_batch(orc, prob, S0, x0)
    ctx = pkg.Context(pkg.default_context().device)
    ctx.set_option("lazy_handover", 1)
    db = pkg.DeviceBatch(prob, S0, x0, ctx=ctx)
    dev = db.S.device
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.current_stream(dev)]
    for rep in range(6):
        s = streams[rep % 3]
        with torch.cuda.stream(s):
            db.solve()                       # no sync in between: the next call moves to another stream
    ctx.sync(streams[5 % 3].cuda_stream)
    torch.cuda.synchronize(dev)
    r = db.results()
    assert_parity(r["z"], r["S"], r["status"], zo, So, sto)
    assert ((r["stats"]["path"] & 64) != 0).all()        # every QP went through the owed big-factor stage


def test_multipliers_are_zero_for_a_qp_that_did_not_converge(pkg, orc):
    """lambda / gamma are written for status > 0 only and every entry point zeroes them first: a solve that runs into the
    iteration limit must not leave the multipliers of an earlier, converged solve of the same DeviceBatch behind"""
    cfg = pkg.CONFIGS["cfg4"]
    prob = pkg.generate_batch(cfg, 16)
    x0, S0, st = pkg.phase1_batch(prob)
    db = pkg.DeviceBatch(prob, S0, x0).want_multipliers()
    db.solve()
    r = db.results()
    assert (r["status"] > 0).all() and np.abs(r["gam"]).max() > 0 and np.abs(r["lam"]).max() > 0
    db.solve(settings=pkg.Settings(maxIter=5))
    r2 = db.results()
    assert (r2["status"] == -6).all()                     # SSQP.jl:271-274
    assert not r2["gam"].any() and not r2["lam"].any()
    z, S, status, detail, lam, gam = pkg.solveQP_batch(prob, S0, x0, settings=pkg.Settings(maxIter=5), want_mult=True)
    assert (status == -6).all() and not gam.any() and not lam.any()


def test_multi_context_entry_over_every_device(pkg, orc):
    """ssqp_solve_batch_multi_f64 with one context per device present (1 on a one-GPU box, 8 on a node) plus a second
    context on device 0, so that the sharded path is exercised even with one card: contiguous blocks, results land in
    the caller's arrays, every problem against the oracle"""
    import torch
    ndev = torch.cuda.device_count()
    cfg = pkg.CONFIGS["cfg4"]
    nprob = 24 * (ndev + 1) + 5                            # (a ragged last block)
    prob = pkg.generate_batch(cfg, nprob, 4242)
    x0, S0, st = pkg.phase1_batch(prob)
    zo, So, sto, _, _ = oracle_batch(orc, prob, S0, x0)
    ctxs = [pkg.Context(d) for d in range(ndev)] + [pkg.Context(0)]
    z, S, status, detail = pkg.solveQP_batch_multi(prob, S0, x0, ctxs)
    assert_parity(z, S, status, zo, So, sto)


def _two_stage_reference(pkg, orc, prob):
    """solveQP(Q) = initQP + the loop (SSQP.jl:224-234) by the host Phase-1 and the oracle's loop"""
    x0, S0, st = pkg.phase1_batch(prob)
    P, N = prob["q"].shape
    z, S, status = x0.copy(), S0.copy(), st.astype(np.int64)
    ok = st == 1
    if ok.any():
        sub = {k: np.ascontiguousarray(v[ok]) for k, v in prob.items()}
        zo, So, sto, _, _ = oracle_batch(orc, sub, S0[ok], x0[ok])
        z[ok], S[ok], status[ok] = zo, So, sto
    return z, S, status, st


@pytest.mark.parametrize("name,nprob", [("cfg4", 160), ("cfg3", 48), ("cfg1", 64), ("cfg2", 8)])
def test_single_launch_solveQP_matches_the_two_stage_path(pkg, orc, name, nprob):
    """ssqp_solve_full_batch_dev_f64: Phase-1 and the loop in one kernel launch per QP -- z, S, status as from the host
    Phase-1 followed by the oracle's loop, the multipliers included"""
    cfg = pkg.CONFIGS[name]
    prob = pkg.generate_batch(cfg, nprob, 20261005)
    zr, Sr, str_, st1 = _two_stage_reference(pkg, orc, prob)
    assert (st1 == 1).all()
    P, N = prob["q"].shape
    db = pkg.DeviceBatch(prob, np.zeros((P, N + cfg.J), dtype=np.int32), np.zeros((P, N))).want_multipliers()
    db.solve_full()
    r = db.results()
    assert_parity(r["z"], r["S"], r["status"], zr, Sr, str_)
    assert ((r["stats"]["path"] & 16) != 0).all()                      # the wavefront kernel's loop ran every QP
    db.solve_full()                                                    # again on the same buffers: nothing stale
    r2 = db.results()
    assert np.array_equal(r2["S"], r["S"]) and np.array_equal(r2["status"], r["status"]) and np.array_equal(r2["z"], r["z"])
    # the two-launch path on the same context gives the same bits (same loop kernel, same vertex)
    db2 = pkg.DeviceBatch(prob, np.zeros((P, N + cfg.J), dtype=np.int32), np.zeros((P, N)))
    db2.phase1()
    db2.solve()
    r3 = db2.results()
    assert np.array_equal(r3["S"], r["S"]) and np.array_equal(r3["status"], r["status"]) and np.array_equal(r3["z"], r["z"])


def test_single_launch_solveQP_mixed_outcomes(pkg, orc):
    """one batch with every way a QP can leave the single launch: solved by the loop; infeasible in Phase-1 (status 0, z = x0,
    Phase-1's S, SSQP.jl:230-232); free variables (left to the workgroup Phase-1 kernel, then handed to the loop at pass 0);
    free variables AND infeasible"""
    cfg = pkg.GenConfig(96, 2, 5, 192, 1e-3, 4.0 / 96, 1.0, 0.1)
    prob = pkg.generate_batch(cfg, 40, 99)
    prob["u"][5:10] = 0.5 / cfg.N                    # the budget row cannot be met
    prob["d"][10:20, :3] = -np.inf                   # free variables
    prob["u"][10:20, :3] = np.inf
    prob["u"][18:20, 3:] = 0.25 / cfg.N              # ... and infeasible: the bounded part cannot reach the budget? (free ones can)
    zr, Sr, str_, st1 = _two_stage_reference(pkg, orc, prob)
    assert (st1[5:10] == 0).all() and (st1[:5] == 1).all() and (st1[10:18] == 1).all()
    P, N = prob["q"].shape
    db = pkg.DeviceBatch(prob, np.zeros((P, N + cfg.J), dtype=np.int32), np.zeros((P, N)))
    db.solve_full()
    r = db.results()
    assert np.array_equal(r["status"], str_), (r["status"], str_)
    assert np.array_equal(r["S"], Sr)
    bad = str_ <= 0
    assert np.array_equal(r["z"][bad], zr[bad])                       # Phase-1's x0, bit for bit
    assert_parity(r["z"][~bad], r["S"][~bad], r["status"][~bad], zr[~bad], Sr[~bad], str_[~bad])
    assert ((r["stats"]["path"][10:18] & 32) != 0).all() or ((r["stats"]["path"][10:18] & 64) != 0).all()   # handed over at pass 0


def test_single_launch_solveQP_refuses_what_it_is_not_built_for(pkg):
    cfg = pkg.GenConfig(64, 2, 12, 128, 1e-3, 0.1, 1.0, 0.1)           # M + J = 14 rows
    prob = pkg.generate_batch(cfg, 4, 1)
    db = pkg.DeviceBatch(prob, np.zeros((4, 64 + 12), dtype=np.int32), np.zeros((4, 64)))
    with pytest.raises(pkg.SSQPError):
        db.solve_full()


@pytest.mark.parametrize("name,nprob,opts", [("cfg4", 128, dict()), ("cfg4", 128, dict(wave_qp_per_cu=8)), ("cfg3", 32, dict())])
def test_lean_builds_without_statistics_give_the_same_results(pkg, orc, name, nprob, opts):
    """a launch with stats = NULL and no trace runs the kernel builds that do not carry the byte / flop accounting
    (-DSSQP_WAVE_LEAN): same z, S, status as the accounting builds and as the oracle -- all three builds of the chain"""
    cfg = pkg.CONFIGS[name]
    prob = pkg.generate_batch(cfg, nprob, 555)
    x0, S0, st = pkg.phase1_batch(prob)
    zo, So, sto, _, _ = oracle_batch(orc, prob, S0, x0)
    ctx = pkg.Context(pkg.default_context().device)
    for k, v in opts.items():
        ctx.set_option(k, v)
    db = pkg.DeviceBatch(prob, S0, x0, ctx=ctx)
    db.solve()
    ra = db.results()
    db.use_stats = False
    db.stats.zero_()
    db.solve()
    rl = db.results()
    assert not rl["stats"]["iters"].any()                      # (the lean launch wrote no statistics)
    assert np.array_equal(rl["z"], ra["z"]) and np.array_equal(rl["S"], ra["S"]) and np.array_equal(rl["status"], ra["status"])
    assert_parity(rl["z"], rl["S"], rl["status"], zo, So, sto)


@pytest.mark.parametrize("shape", [(48, 3, 10), (64, 2, 30), (96, 4, 60), (80, 4, 80), (40, 3, 84)])
@pytest.mark.parametrize("kind", ["plain", "some_free", "infeasible"])
def test_phase1_many_rows_build(pkg, orc, shape, kind):
    """the many-rows build of the workgroup Phase-1 kernel (M + J > 12: 512 threads, sparsity-aware inv(lu(B)) with the
    logical row map, listed steps of the Y.c refresh, staged xb) at 13, 32, 64, 84 and 87 rows (the most that fit in LDS) -- one and two 64-lane chunks
    per column, three and six lanes per column of the inverse -- bit for bit the host stage's (x0, S0, status), which the
    CPU suite pins to the oracle"""
    N, M, J = shape
    ub = 0.0 if kind == "some_free" else 4.0 / N
    cfg = pkg.GenConfig(N, M, J, 2 * N, 1e-3, ub, 1.0, 0.2)
    prob = pkg.generate_batch(cfg, 6, 1000 + N + J)
    if kind == "infeasible":
        prob["u"][:] = 0.5 / N
    elif kind == "some_free":
        prob["d"][:, ::7] = -np.inf
        prob["u"][:, ::7] = np.inf
        prob["u"][:, 1::7] = 4.0 / N
        prob["d"][:, 1::7] = 0.0
        prob["u"][:, 2::7] = 4.0 / N
    xh, Sh, sth = pkg.phase1_batch(prob)
    xo, So, sto = orc.initQP_batch(prob["A"], prob["G"], prob["b"], prob["g"], prob["d"], prob["u"])
    assert np.array_equal(sth, sto) and np.array_equal(Sh, So) and np.array_equal(xh, xo)
    P = prob["q"].shape[0]
    db = pkg.DeviceBatch(prob, np.zeros((P, N + J), dtype=np.int32), np.zeros((P, N)))
    st = db.phase1()
    db.torch.cuda.synchronize()
    assert np.array_equal(st.cpu().numpy(), sth), (shape, kind, st.cpu().numpy(), sth)
    assert np.array_equal(db.S0.cpu().numpy(), Sh) and np.array_equal(db.x0.cpu().numpy(), xh), (shape, kind)
    if kind == "infeasible":
        assert (sth == 0).all()


def test_phase1_many_rows_random_shapes(pkg):
    """forty random shapes with 13 .. 87 rows (N 16 .. 160, equality rows 1 .. 6, the rest inequalities; bounds tight, loose, partly
    free, partly infeasible): the sparsity-aware elimination meets runs of exchanges before and after steps with a nonzero L
    column, moved rows on both sides of lane 64, bases with few and with many dense columns -- (x0, S0, status) bit for bit the
    host stage's on every problem"""
    rng = np.random.default_rng(20261005)
    seen = set()
    for trial in range(40):
        M0 = int(rng.integers(13, 88))
        M = int(rng.integers(1, 7))
        J = M0 - M
        N = int(rng.integers(16, 161))
        kind = ["tight", "loose", "free", "infeasible"][trial % 4]
        ub = {"tight": 2.5 / N, "loose": 20.0 / N, "free": 0.0, "infeasible": 3.0 / N}[kind]
        cfg = pkg.GenConfig(N, M, J, 2 * N, 1e-3, ub, float(rng.uniform(0.8, 1.3)), 0.2)
        prob = pkg.generate_batch(cfg, 4, int(rng.integers(1, 2 ** 31)))
        if kind == "free":
            free = rng.random(N) < 0.2
            prob["d"][:, free] = -np.inf
            prob["u"][:, free] = np.inf
            prob["u"][:, ~free] = 6.0 / N
        if kind == "infeasible":
            prob["u"][2:] = 0.5 / N
        xh, Sh, sth = pkg.phase1_batch(prob)
        db = pkg.DeviceBatch(prob, np.zeros((4, N + J), dtype=np.int32), np.zeros((4, N)))
        st = db.phase1()
        db.torch.cuda.synchronize()
        assert np.array_equal(st.cpu().numpy(), sth), (trial, N, M, J, kind, st.cpu().numpy(), sth)
        assert np.array_equal(db.S0.cpu().numpy(), Sh), (trial, N, M, J, kind)
        assert np.array_equal(db.x0.cpu().numpy(), xh), (trial, N, M, J, kind)
        seen.update(sth.tolist())
    assert {0, 1} <= seen


def test_phase1_many_rows_wide_lp(pkg):
    """N1 = N + J + M + J > 8 x 512 columns: the workgroup-wide list of the columns at a nonzero bound takes a second round
    (csrc/ssqp_phase1.hip, compact_columns_wg), the Y.c refresh more than two column blocks -- bit-identical to the host stage"""
    N, M, J = 4400, 2, 14
    cfg = pkg.GenConfig(N, M, J, 64, 1e-3, 30.0 / N, 1.0, 0.2)
    prob = pkg.generate_batch(cfg, 2, 31337)
    xh, Sh, sth = pkg.phase1_batch(prob)
    assert (sth == 1).all()
    db = pkg.DeviceBatch(prob, np.zeros((2, N + J), dtype=np.int32), np.zeros((2, N)))
    st = db.phase1()
    db.torch.cuda.synchronize()
    assert np.array_equal(st.cpu().numpy(), sth)
    assert np.array_equal(db.S0.cpu().numpy(), Sh) and np.array_equal(db.x0.cpu().numpy(), xh)
    assert (xh != 0).sum(axis=1).min() >= 30          # (the list of columns at a nonzero bound is not trivial)
