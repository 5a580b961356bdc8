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
