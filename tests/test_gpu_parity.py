"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bar (BASELINE.json north_star): status vectors bit-exact, iteration counts equal,
z within 1e-10 relative (Float64)."""
import glob
import os

import numpy as np
import pytest

# SSQP_TEST_SEED=<n> shifts the seeds of the randomised shape tests (extra sweeps on a GPU box)
SEED_SHIFT = int(os.environ.get("SSQP_TEST_SEED", "0"))

from conftest import assert_parity, colmajor, oracle_batch

pytestmark = pytest.mark.gpu


def run_cfg(pkg, orc, cfg, nprob, seed0=None, rtol=1e-10, opts=None, both=True):
    """HIP path vs oracle on one synthetic family.  `opts` = context options for the run (ssqp_ctx_set_option).
    With the default options a shape the wavefront-per-QP kernel takes is ALSO run through the workgroup kernel
    alone (wave_kernel=0): both product kernels must match the oracle.  Returns the stats of the first run."""
    prob = pkg.generate_batch(cfg, nprob, seed0 if seed0 is not None else pkg.BASE_SEED)
    x0, S0, st = pkg.phase1_batch(prob)
    assert (st == 1).all()
    zo, So, sto, deto, _ = oracle_batch(orc, prob, S0, x0)
    assert (sto > 0).all()
    ctx = pkg.default_context()
    runs = [dict(opts or {})]
    if opts is None and both and cfg.N % 2 == 0 and cfg.N <= 512 and cfg.M + cfg.J <= 11:
        runs.append(dict(wave_kernel=0))
        runs.append(dict(wave_qp_per_cu=8))     # the two-wavefronts-per-SIMD build of the wavefront kernel
    first = None
    for o in runs:
        with ctx.options(**o):
            z, S, status, detail, stats = pkg.solveQP_batch(prob, S0, x0, want_stats=True)
        rel = assert_parity(z, S, status, zo, So, sto, rtol)
        assert (detail == 0).all()
        assert np.array_equal(stats["iters"], sto)
        if first is None:
            first = (rel, stats)
    return first


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
    assert ((stats["path"] & 2) == 0).all()      # never left the LDS arena (bit 2 = incremental engine used)


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
    assert ((stats["path"] & 2) == 0).all()      # never left the LDS arena (bit 2 = incremental engine used)


def test_cfg3_n256_large_k_global_arena(pkg, orc):
    """K grows to ~220: the kept factor outgrows LDS and migrates to the workgroup's global arena (path bit 8)"""
    rel, stats = run_cfg(pkg, orc, pkg.CONFIGS["cfg3"], 8, opts=dict(wave_kernel=0))
    assert stats["max_k"].max() > 180
    assert ((stats["path"] & 8) != 0).all()
    # default routing: the wavefront kernel starts every QP and its big-factor build takes over when K outgrows the
    # first build's factor (bit 64); nothing reaches the workgroup kernel (bit 32)
    rel, stats = run_cfg(pkg, orc, pkg.CONFIGS["cfg3"], 8, both=False)
    assert stats["max_k"].max() > 180
    assert ((stats["path"] & 16) != 0).all() and ((stats["path"] & 64) != 0).all() and ((stats["path"] & 32) == 0).all()
    # from-scratch factorisation out of the global arena
    rel, stats = run_cfg(pkg, orc, pkg.CONFIGS["cfg3"], 2, opts=dict(incremental=0))
    assert ((stats["path"] & 2) != 0).all()


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


# ---------------------------------------------------------------- golden fixtures through the C ABI
import glob
import os

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_fixture(pkg, path):
    d = np.load(path)
    c = {k: d[k] for k in d.files}
    Q = pkg.QP.inner(c["V"], c["A"], c["G"], c["q"], c["b"], c["g"], c["d"], c["u"])
    if int(c["phase1_status"]) != 1:
        z, S, status = pkg.solveQP(Q)                 # full path: Phase-1 on the host reports infeasibility
        assert status == 0
        return
    z, S, status = pkg.solveQP(Q)                     # solveQP(Q): Phase-1 + loop
    assert status == int(c["status"]) and np.array_equal(S, c["S"])
    np.testing.assert_allclose(z, c["z"], rtol=0, atol=1e-10 * max(1.0, np.abs(c["z"]).max()))
    S2 = c["S0"].astype(np.int32).copy()              # solveQP(Q, S, x0): S is mutated in place and returned
    z2, Sret, status2 = pkg.solveQP(Q, S2, c["x0"])
    assert Sret is S2 and np.array_equal(S2, c["S"]) and status2 == int(c["status"])
    # per-iteration trace (K, W, event kind, switched id) identical to the oracle's
    prob = dict(V=c["V"][None], A=np.ascontiguousarray(c["A"].T)[None], G=np.ascontiguousarray(c["G"].T)[None],
                q=c["q"][None], b=c["b"][None], g=c["g"][None], d=c["d"][None], u=c["u"][None])
    db = pkg.DeviceBatch(prob, c["S0"][None], c["x0"][None], ntrace=4096)
    db.solve()
    r = db.results()
    assert np.array_equal(r["trace"][0][:status], c["trace"])


def test_status_codes_on_gpu(pkg):
    d = np.load([p for p in GOLDEN if p.endswith("box20_j3.npz")][0])
    c = {k: d[k] for k in d.files}
    args = (c["A"], c["G"], c["q"], c["b"], c["g"], c["d"], c["u"])
    Q = pkg.QP.inner(c["V"], *args, mc=-70)                                   # SSQP.jl:226-228
    z, S, status, det = pkg.solveQP(Q, return_detail=True)
    assert status == -1 and det == 4 and (S == pkg.DN).all() and len(S) == 20 and (z == 0).all()
    Q = pkg.QP.inner(c["V"], *args)
    z, S, status = pkg.solveQP(Q, c["S0"].astype(np.int32).copy(), c["x0"], settings=pkg.Settings(maxIter=5))
    assert status == -6                                                       # SSQP.jl:272-273
    Qbad = pkg.QP.inner(c["V"] - 10.0 * np.eye(20), *args)                    # cholesky(V[F,F]) would throw
    z, S, status, det = pkg.solveQP(Qbad, c["S0"].astype(np.int32).copy(), c["x0"], return_detail=True)
    assert status == -1 and det == 1


# ---------------------------------------------------------------- full-size batches: properties
def test_full_batch_properties_cfg4(pkg, orc):
    """1024 x N=512 (BASELINE.json configs[3] per GPU): everything converges, a sample is bit-exact against the
    oracle and passes the independent KKT check, and re-solving from the solution is a fixed point."""
    from kkt import assert_kkt
    cfg = pkg.CONFIGS["cfg4"]
    prob = pkg.generate_batch(cfg, 1024)
    x0, S0, st = pkg.phase1_batch(prob)
    z, S, status, detail, stats = pkg.solveQP_batch(prob, S0, x0, want_stats=True)
    assert (status > 0).all() and (detail == 0).all()
    # feasibility; polishSz! snaps variables within tol = 2^-26 of a bound, so the budget holds to ~N*tol
    assert np.abs(z.sum(axis=1) - 1.0).max() < 1e-6 and z.min() >= 0.0 and z.max() <= cfg.ub
    sel = np.arange(0, 1024, 16)
    sub = {k: v[sel] for k, v in prob.items()}
    zo, So, sto, _, _ = oracle_batch(orc, sub, S0[sel], x0[sel])
    assert_parity(z[sel], S[sel], status[sel], zo, So, sto)
    for p in range(1024):   # independent of the restatement: every one of the 1024 solutions is a KKT point
        assert_kkt(prob["V"][p], colmajor(prob["A"][p], cfg.M), colmajor(prob["G"][p], cfg.J), prob["q"][p],
                   prob["b"][p], prob["g"][p], prob["d"][p], prob["u"][p], z[p], S[p])
    # idempotence: warm start at the optimum leaves S unchanged and stops after one pass
    z2, S2, status2, _ = pkg.solveQP_batch(prob, S, z)
    # (polishSz! may relabel a near-bound IN variable, so the warm start can need a couple of passes)
    assert np.array_equal(S2, S) and (status2 >= 1).all() and (status2 <= 3).all() and np.abs(z2 - z).max() < 1e-9
    # the dense formulation of the gamma pass (every column of V read, as SSQP.jl:352) takes the same decisions
    with pkg.default_context().options(dense_gamma=1):
        z3, S3, status3, _ = pkg.solveQP_batch(prob, S0, x0)
    assert np.array_equal(S3, S) and np.array_equal(status3, status) and np.abs(z3 - z).max() < 1e-12


def test_cfg5_rank_deficient_n2048(pkg, orc):
    """BASELINE.json configs[4]: N=2048, M=8, J=64, V = X'X/T with T=1024 (rank 1024, no ridge).  The reference
    only works while V[F,F] stays PD; on this instance it does (K <= 136)."""
    cfg = pkg.CONFIGS["cfg5"]
    rel, stats = run_cfg(pkg, orc, cfg, 1)
    assert stats["max_k"][0] < 1024


def test_batch_larger_than_grid(pkg, orc):
    """more problems than resident workgroups: the work queue hands several problems to one workgroup"""
    cfg = pkg.GenConfig(96, 1, 4, 192, 1e-3, 0.08, 1.0, 0.1)
    run_cfg(pkg, orc, cfg, 1500)


def test_device_generator_is_bit_identical(pkg):
    """ssqp_generate_V_dev makes exactly the host generator's V (same counter stream, same summation order)."""
    for cfg in (pkg.GenConfig(48, 1, 2, 70, 1e-3, 0.1, 1.0, 0.1), pkg.GenConfig(130, 2, 3, 64, 0.0, 0.05, 1.0, 0.1)):
        host = pkg.generate_batch(cfg, 3, 4242)
        batch, prob, x0, S0 = pkg.DeviceBatch.generated(cfg, 3, 4242)
        assert np.array_equal(batch.t["V"].cpu().numpy(), host["V"])
        for k in "AGqbgdu":
            assert np.array_equal(prob[k], host[k])


def test_incremental_and_from_scratch_agree(pkg, orc):
    """the kept-factor engine (default) and the from-scratch factorisation (option incremental=0) take the same
    decisions as the oracle"""
    cfg = pkg.GenConfig(160, 1, 5, 320, 1e-3, 0.06, 0.98, 0.1)
    for inc in (1, 0):
        rel, stats = run_cfg(pkg, orc, cfg, 48, opts=dict(incremental=inc, wave_kernel=0))
        assert (((stats["path"] & 4) != 0).all()) == (inc == 1)
    rel, stats = run_cfg(pkg, orc, cfg, 48, both=False)      # default routing: the wavefront kernel
    assert ((stats["path"] & 16) != 0).all()


def _mutate(prob, rng, kind):
    """variations the plain generator does not produce"""
    P, N = prob["q"].shape
    if kind == "lower_bounds":          # nonzero lower bounds: bound variables at DN carry weight in V[:,B] zB
        prob["d"][:] = rng.uniform(0.0, 0.3 / N, size=(P, N))
        prob["u"][:] = prob["d"] + rng.uniform(0.5 / N, 4.0 / N, size=(P, N)) + 1.0 / N
    elif kind == "negative_lower":      # shorting allowed
        prob["d"][:] = -rng.uniform(0.0, 1.0 / N, size=(P, N))
    elif kind == "dup_rows" and prob["G"].shape[2] >= 2:   # dependent inequality rows -> rank filter purges
        J = prob["G"].shape[2]
        G = prob["G"].reshape(P, N, J)
        G[:, :, 1] = G[:, :, 0]
        prob["g"][:, 1] = prob["g"][:, 0]
    elif kind == "some_free":           # a few variables without bounds
        prob["d"][:, :3] = -np.inf
        prob["u"][:, :3] = np.inf
    elif kind == "big_q":
        prob["q"][:] *= 5.0
    return prob


@pytest.mark.parametrize("kind", ["plain", "lower_bounds", "negative_lower", "dup_rows", "some_free", "big_q"])
def test_random_shapes_and_bounds(pkg, orc, kind):
    """many small random problems of mixed shapes; every one must match the oracle decision for decision"""
    import zlib
    rng = np.random.default_rng(zlib.crc32(kind.encode()) + SEED_SHIFT)
    shapes = [(24, 1, 0), (40, 1, 3), (57, 2, 4), (64, 1, 6), (96, 3, 8), (130, 1, 10), (150, 2, 5), (200, 1, 2)]
    _check_shapes(pkg, orc, kind, rng, shapes, 12, 40)


@pytest.mark.parametrize("kind", ["plain", "lower_bounds", "dup_rows"])
def test_many_inequalities_and_wide_shapes(pkg, orc, kind):
    """more constraint rows than the cached-row batch holds (M+J > 12: full row sweeps, possibly more than 12 active
    rows -> the workgroup rank filter and the from-scratch factorisation), and N up to 512 with up to 12 rows"""
    import zlib
    rng = np.random.default_rng(zlib.crc32(("wide" + kind).encode()) + SEED_SHIFT)
    shapes = [(80, 1, 14), (120, 2, 20), (96, 1, 12), (256, 1, 11), (384, 2, 9), (512, 1, 6)]
    _check_shapes(pkg, orc, kind, rng, shapes, 6, 20)


def _check_shapes(pkg, orc, kind, rng, shapes, nper, need):
    total = 0
    for (N, M, J) in shapes:
        ub = rng.uniform(2.0, 6.0) / N if kind != "some_free" else 0.0
        cfg = pkg.GenConfig(N, M, J, 2 * N, 1e-3, ub, rng.uniform(0.93, 1.1), rng.uniform(0.0, 0.3))
        prob = pkg.generate_batch(cfg, nper, int(rng.integers(1, 2 ** 31)))
        prob = _mutate(prob, rng, kind)
        x0, S0, st = pkg.phase1_batch(prob)
        ok = st == 1
        if not ok.any():
            continue
        sub = {k: np.ascontiguousarray(v[ok]) for k, v in prob.items()}
        z, S, status, detail = pkg.solveQP_batch(sub, S0[ok], x0[ok])
        if N % 2 == 0 and N <= 512 and M + J <= 11:      # ... and the same through the eight-per-CU wavefront kernel
            with pkg.default_context().options(wave_qp_per_cu=8):
                z8, S8, status8, _ = pkg.solveQP_batch(sub, S0[ok], x0[ok])
            # (same source, two register budgets: the compiler may contract a*b+c differently, so z agrees to rounding)
            assert np.array_equal(status8, status) and np.array_equal(S8, S), (kind, N, M, J)
            fin8 = np.isfinite(z).all(axis=1)
            assert np.abs(z8[fin8] - z[fin8]).max() <= 1e-12 * max(1.0, np.abs(z[fin8]).max()) if fin8.any() else True
        zo, So, sto, deto, _ = oracle_batch(orc, sub, S0[ok], x0[ok])
        conv = sto > 0
        assert np.array_equal(status, sto), (kind, N, M, J, status, sto)
        assert np.array_equal(S[conv], So[conv])
        fin = np.isfinite(zo).all(axis=1) & conv
        scale = np.maximum(np.abs(zo[fin]).max(axis=1), 1e-300)
        assert (np.abs(z[fin] - zo[fin]).max(axis=1) / scale).max() < 1e-10 if fin.any() else True
        # the check that shares nothing with the restatement: every converged solution satisfies the KKT conditions
        # (a strictly convex QP has one optimum).  Free variables without bounds are skipped by the verifier's
        # bound tests automatically (+-Inf bounds).
        from kkt import assert_kkt
        for p in np.flatnonzero(fin):
            assert_kkt(sub["V"][p], colmajor(sub["A"][p], M), colmajor(sub["G"][p], J), sub["q"][p], sub["b"][p],
                       sub["g"][p], sub["d"][p], sub["u"][p], z[p], S[p], eps=2e-7)
        total += int(conv.sum())
    assert total > need


def test_shared_v_batch_efficient_frontier_style(pkg, orc):
    """One V, A, G, d, u for the whole batch, only q differs per problem -- QP(P, q, L) of the reference
    (src/types.jl:303-319) swept over L: the strided entry point with stride 0 on the shared arrays."""
    cfg = pkg.GenConfig(128, 1, 4, 256, 1e-3, 0.05, 1.05, 1.0)
    base = pkg.generate_batch(cfg, 1, 31337)
    Ls = np.linspace(0.0, 2.0, 24)
    P = len(Ls)
    q = np.ascontiguousarray(Ls[:, None] * base["q"][0][None, :])          # q_L = -L * mu
    full = {k: np.ascontiguousarray(np.repeat(base[k], P, axis=0)) for k in "VAGbgdu"}
    full["q"] = q
    x0, S0, st = pkg.phase1_batch(full)
    assert (st == 1).all()
    shared = dict(base)
    shared["q"] = q
    db = pkg.DeviceBatch(shared, S0, x0)           # V, A, G, b, g, d, u have leading dimension 1 -> stride 0
    db.solve()
    r = db.results()
    zo, So, sto, _, _ = oracle_batch(orc, full, S0, x0)
    assert_parity(r["z"], r["S"], r["status"], zo, So, sto)
    assert len(set(map(tuple, r["S"]))) > 3       # the frontier really moves through different active sets


def test_launch_lanes_overlap_and_agree(pkg, orc):
    """three launch lanes (one context and one HIP stream each, shared inputs, own outputs -- what bench.py uses to
    overlap the drain of one batch with the start of the next): overlapping launches give the decisions of a
    serial launch, and those of the oracle"""
    import torch
    cfg = pkg.GenConfig(96, 1, 6, 192, 1e-3, 0.08, 1.02, 0.15)
    b0, prob, x0, S0 = pkg.DeviceBatch.generated(cfg, 600, 777)
    dev = b0.S.device
    lanes = [(b0, torch.cuda.current_stream(dev))]
    for _ in range(2):
        lanes.append((b0.twin(pkg.Context(dev.index)), torch.cuda.Stream(dev)))
    for rep in range(2):
        for b, st in lanes:
            with torch.cuda.stream(st):
                b.solve()
    torch.cuda.synchronize(dev)
    ref = lanes[0][0].results()
    for b, _ in lanes[1:]:
        r = b.results()
        assert np.array_equal(r["S"], ref["S"]) and np.array_equal(r["status"], ref["status"])
        assert np.array_equal(r["z"], ref["z"])
    full = dict(prob)
    full["V"] = b0.t["V"].cpu().numpy()
    zo, So, sto, _, _ = oracle_batch(orc, full, S0, x0)
    assert_parity(ref["z"], ref["S"], ref["status"], zo, So, sto)


# ---------------------------------------------------------------- paths no BASELINE config reaches by itself
def _oracle_fast(orc, prob, S0, x0):
    """the oracle with LAPACK arithmetic when scipy's OpenBLAS can be bound (large K: the port's loops take minutes)"""
    return orc.solveQP_warm_batch(prob["V"], prob["A"], prob["G"], prob["q"], prob["b"], prob["g"], prob["d"],
                                  prob["u"], S0, x0, lapack=orc.lapack_available())


def test_free_set_beyond_256_rows(pkg, orc):
    """V ~ I/2 without upper bounds: nearly every variable ends up free, K grows past the 256 rows the kept factor
    holds -- the workgroup kernel's from-scratch / global-arena / MFMA-panel path as the PRODUCT path, reached
    through the hand-over chain wavefront kernel -> kept factor in LDS -> global arena -> from scratch."""
    cfg = pkg.GenConfig(320, 1, 4, 640, 0.5, 0.0, 1.05, 0.0)
    prob = pkg.generate_batch(cfg, 4, 90210)
    x0, S0, st = pkg.phase1_batch(prob)
    assert (st == 1).all()
    zo, So, sto, _, _ = _oracle_fast(orc, prob, S0, x0)
    assert (sto > 0).all()
    ctx = pkg.default_context()
    for opts in (dict(), dict(wave_kernel=0)):
        with ctx.options(**opts):
            z, S, status, detail, stats = pkg.solveQP_batch(prob, S0, x0, want_stats=True)
        assert_parity(z, S, status, zo, So, sto)
        assert stats["max_k"].min() > 256, stats["max_k"]
        assert ((stats["path"] & 2) != 0).all()                  # the global arena was used
        assert ((stats["path"] & 8) != 0).all()                  # ... after the kept factor had migrated there
        if not opts:
            assert ((stats["path"] & 112) == 112).all()          # wavefront kernel -> its big-factor build -> workgroup kernel


def test_k0_start_n512_freeK(pkg, orc):
    """warm start with NO free variable at N = 512 (S = DN everywhere, z = d): the first pass is freeK!
    (SSQP.jl:35-59) on the full N x N product, which releases every variable whose gradient allows"""
    cfg = pkg.CONFIGS["cfg4"]
    prob = pkg.generate_batch(cfg, 3, 4711)
    P, N, J = 3, cfg.N, cfg.J
    S0 = np.full((P, N + J), pkg.DN, dtype=np.int32)
    S0[:, N:] = pkg.OE
    x0 = prob["d"].copy()
    zo, So, sto, _, _ = _oracle_fast(orc, prob, S0, x0)
    ctx = pkg.default_context()
    for opts in (dict(), dict(wave_kernel=0)):
        with ctx.options(**opts):
            db = pkg.DeviceBatch(prob, S0, x0, ntrace=8)
            db.solve()
            r = db.results()
        assert np.array_equal(r["status"], sto), (r["status"], sto)
        conv = sto > 0
        assert np.array_equal(r["S"][conv], So[conv])
        assert (r["trace"][:, 0, 2] == 0).all() and (r["trace"][:, 0, 0] == 0).all()   # pass 1: a K == 0 pass
        if conv.any():
            scale = np.maximum(np.abs(zo[conv]).max(axis=1), 1e-300)
            assert (np.abs(r["z"][conv] - zo[conv]).max(axis=1) / scale).max() < 1e-10


def test_cfg4_j0_family(pkg, orc):
    """the J = 0 variant of the headline family"""
    rel, stats = run_cfg(pkg, orc, pkg.CONFIGS["cfg4_j0"], 24)
    assert ((stats["path"] & 16) != 0).all()


def test_cfg3_full_batch_1024(pkg, orc):
    """BASELINE.json configs[2] at full size: 1024 x N=256, K up to ~230 (wavefront kernel -> its big-factor build), on
    every start: four- and eight-per-CU first stage, and the big-factor build from the first pass"""
    for opts in (None, dict(wave_qp_per_cu=8), dict(wave_kernel=2)):
        rel, stats = run_cfg(pkg, orc, pkg.CONFIGS["cfg3"], 1024, both=False, opts=opts)
        assert stats["max_k"].max() > 200
        assert ((stats["path"] & 32) == 0).all()         # nothing needed the workgroup kernel
        assert (((stats["path"] & 64) != 0).all()) == (not opts or "wave_kernel" not in opts)


@pytest.mark.parametrize("N", [768, 601, 1500])
def test_wide_n_kernels(pkg, orc, N):
    """N in 513..1024 even (ssqp_solve_kernel<3,1>: eight accumulator slots per lane), odd N > 512 (scalar loads
    with the maximum number of per-thread slots) and N = 1500 with fourteen inequality rows (ssqp_solve_kernel<4,1>; the
    eight-loads-deep row streams of the ratio test and the E-row sweep end in a partial second batch)"""
    cfg = pkg.GenConfig(N, 1, 3 if N < 1100 else 14, 2 * N if N < 1100 else N, 1e-3, 4.0 / N, 1.05, 0.1)
    run_cfg(pkg, orc, cfg, 2, seed0=1234 + N)


def test_eight_per_cu_build_parks_second_row_slot(pkg, orc):
    """a batch above 4 QPs per CU goes to the eight-per-CU build by default; here every QP ends with 70-82 free variables:
    rows 64.. live in the second row slot, which that build keeps in global scratch between passes -- same results as
    the four-per-CU build, nothing handed over"""
    import torch
    ctx = pkg.default_context()
    ncu = torch.cuda.get_device_properties(ctx.device).multi_processor_count
    nprob = 4 * ncu + 40
    cfg = pkg.GenConfig(88, 1, 0, 176, 1e-3, 0.0, 1.2, 0.0)
    prob = pkg.generate_batch(cfg, nprob, 4242)
    x0, S0, st = pkg.phase1_batch(prob)
    assert (st == 1).all()
    z1, S1, st1, d1, stats1 = pkg.solveQP_batch(prob, S0, x0, want_stats=True)
    with ctx.options(wave_qp_per_cu=4):
        z2, S2, st2, d2, stats2 = pkg.solveQP_batch(prob, S0, x0, want_stats=True)
    for stats in (stats1, stats2):
        assert ((stats["path"] & 48) == 16).all()   # wavefront kernel to the end
        assert stats["max_k"].min() > 63 and stats["max_k"].max() <= 88
    assert np.array_equal(S1, S2) and np.array_equal(st1, st2)
    assert np.allclose(z1, z2, rtol=1e-12, atol=1e-14)
    sub = {k: np.ascontiguousarray(v[:48]) for k, v in prob.items()}
    zo, So, sto, _, _ = oracle_batch(orc, sub, S0[:48], x0[:48])
    assert_parity(z1[:48], S1[:48], st1[:48], zo, So, sto)
    assert_parity(z2[:48], S2[:48], st2[:48], zo, So, sto)


@pytest.mark.parametrize("shape", ["deep", "boundary"])
def test_second_row_slot_with_active_inequalities(pkg, orc, shape):
    """free sets that live in the second row slot of the wavefront kernel: "deep" ends with 105-120 free variables and
    ~6 active inequalities (the eight-per-CU build finishes it with the parked slot, the four-per-CU build hands over at
    its 92 rows); "boundary" ends with 54-77 and upper bounds that block, so appends AND deletes cross row 64 in both
    directions.  Every build against the oracle."""
    if shape == "deep":
        cfg = pkg.GenConfig(128, 1, 8, 256, 1e-3, 0.0, 0.98, 0.0)
    else:
        cfg = pkg.GenConfig(144, 1, 5, 288, 1e-3, 0.03, 1.0, 0.02)
    prob = pkg.generate_batch(cfg, 48, 777 + SEED_SHIFT)
    x0, S0, st = pkg.phase1_batch(prob)
    ok = st == 1
    assert ok.sum() >= 24
    sub = {k: np.ascontiguousarray(v[ok]) for k, v in prob.items()}
    zo, So, sto, _, _ = oracle_batch(orc, sub, S0[ok], x0[ok])
    assert (sto > 0).all()
    ctx = pkg.default_context()
    for opts in (dict(wave_qp_per_cu=8), dict(wave_qp_per_cu=4), dict(wave_kernel=0)):
        with ctx.options(**opts):
            z, S, status, detail, stats = pkg.solveQP_batch(sub, S0[ok], x0[ok], want_stats=True)
        assert_parity(z, S, status, zo, So, sto)
        if shape == "deep" and SEED_SHIFT == 0:
            assert stats["max_k"].min() > 100 and stats["max_k"].max() <= 127
            if opts.get("wave_qp_per_cu") == 8:
                assert ((stats["path"] & 48) == 16).all()    # the wavefront kernel to the end
            if opts.get("wave_qp_per_cu") == 4:
                assert ((stats["path"] & 112) == 80).all()   # beyond 92 rows: continued in the big-factor build
        if shape == "boundary" and SEED_SHIFT == 0 and "wave_kernel" not in opts:
            assert (stats["max_k"] > 64).any() and ((stats["path"] & 32) == 0).all()


def test_two_contexts_one_device_agree_with_one(pkg, orc):
    """ssqp_solve_batch_multi_f64: contiguous blocks over two contexts (here both on device 0, each with its own
    host thread, stream and workspace) give exactly the single-context results"""
    cfg = pkg.GenConfig(96, 1, 5, 192, 1e-3, 0.07, 1.03, 0.1)
    prob = pkg.generate_batch(cfg, 257, 2024)      # (odd count: the last block is shorter)
    x0, S0, st = pkg.phase1_batch(prob)
    assert (st == 1).all()
    z1, S1, st1, d1 = pkg.solveQP_batch(prob, S0, x0)
    dev = pkg.default_context().device
    ctxs = [pkg.Context(dev), pkg.Context(dev)]
    z2, S2, st2, d2 = pkg.solveQP_batch_multi(prob, S0, x0, ctxs)
    assert np.array_equal(S1, S2) and np.array_equal(st1, st2) and np.array_equal(z1, z2)
    z3, S3, st3, d3 = pkg.solveQP_batch_multi(prob, S0, x0, ctxs + [pkg.Context(dev)])
    assert np.array_equal(S1, S3) and np.array_equal(st1, st3) and np.array_equal(z1, z3)
    zo, So, sto, _, _ = oracle_batch(orc, prob, S0, x0)
    assert_parity(z2, S2, st2, zo, So, sto)


def test_long_runs_refresh_caches(pkg, orc):
    """many status switches at nonzero bounds (lower bounds > 0): hq and bEall follow by one column per switch and are
    re-evaluated from (z, S) every 64 switches -- the decisions stay those of the oracle over several hundred passes"""
    rng = np.random.default_rng(99 + SEED_SHIFT)
    cfg = pkg.GenConfig(256, 1, 6, 512, 1e-3, 5.0 / 256, 1.0, 0.2)
    prob = pkg.generate_batch(cfg, 16, 555)
    prob["d"][:] = rng.uniform(0.0, 0.3 / 256, size=prob["d"].shape)
    x0, S0, st = pkg.phase1_batch(prob)
    ok = st == 1
    assert ok.sum() >= 8
    sub = {k: np.ascontiguousarray(v[ok]) for k, v in prob.items()}
    zo, So, sto, _, _ = oracle_batch(orc, sub, S0[ok], x0[ok])
    assert (sto > 150).any()
    ctx = pkg.default_context()
    for opts in (dict(), dict(wave_kernel=0), dict(wave_qp_per_cu=8)):
        with ctx.options(**opts):
            z, S, status, detail = pkg.solveQP_batch(sub, S0[ok], x0[ok])
        assert_parity(z, S, status, zo, So, sto)


# ---------------------------------------------------------------- Phase-1 on the GPU (ssqp_phase1_batch_dev_f64)
def _phase1_gpu(pkg, prob):
    P, N = prob["q"].shape
    db = pkg.DeviceBatch(prob, np.zeros((P, N + prob["g"].shape[1]), dtype=np.int32), np.zeros((P, N)))
    st = db.phase1()
    db.torch.cuda.synchronize()
    return db.x0.cpu().numpy(), db.S0.cpu().numpy(), st.cpu().numpy()


@pytest.mark.parametrize("name,nprob", [("cfg1", 32), ("cfg2", 8), ("cfg4", 256), ("cfg3", 64), ("cfg5", 1)])
def test_phase1_gpu_bit_identical_to_host(pkg, orc, name, nprob):
    """initQP + cDantzigLP (SSQP.jl:461-560, Simplex.jl:445-615) on the GPU: the vertex (x0, S0) and the status are
    bit for bit those of the host C++ stage and of the oracle -- so the loop that follows runs the same passes"""
    cfg = pkg.CONFIGS[name]
    prob = pkg.generate_batch(cfg, nprob)
    xh, Sh, sth = pkg.phase1_batch(prob)
    xo, So, sto = orc.initQP_batch(prob["A"], prob["G"], prob["b"], prob["g"], prob["d"], prob["u"])
    xg, Sg, stg = _phase1_gpu(pkg, prob)
    assert np.array_equal(stg, sth) and np.array_equal(stg, sto)
    assert np.array_equal(Sg, Sh) and np.array_equal(Sg, So)
    assert np.array_equal(xg, xh) and np.array_equal(xg, xo)


@pytest.mark.parametrize("kind", ["plain", "lower_bounds", "negative_lower", "dup_rows", "some_free", "infeasible"])
def test_phase1_gpu_variations(pkg, orc, kind):
    import zlib
    rng = np.random.default_rng(zlib.crc32(("p1" + kind).encode()) + SEED_SHIFT)
    for (N, M, J) in [(24, 1, 0), (40, 1, 3), (57, 2, 4), (96, 3, 8), (200, 1, 2), (128, 2, 20)]:
        ub = rng.uniform(2.0, 6.0) / N if kind != "some_free" else 0.0
        cfg = pkg.GenConfig(N, M, J, 2 * N, 1e-3, ub, rng.uniform(0.93, 1.1), rng.uniform(0.0, 0.3))
        prob = pkg.generate_batch(cfg, 10, int(rng.integers(1, 2 ** 31)))
        if kind == "infeasible":
            prob["u"][:] = 0.5 / N               # the budget cannot be met: sum(u) = 1/2 < 1
        else:
            prob = _mutate(prob, rng, kind)
        xh, Sh, sth = pkg.phase1_batch(prob)
        xg, Sg, stg = _phase1_gpu(pkg, prob)
        assert np.array_equal(stg, sth), (kind, N, M, J, stg, sth)
        assert np.array_equal(Sg, Sh) and np.array_equal(xg, xh), (kind, N, M, J)
        if kind == "infeasible":
            assert (sth == 0).all()


def test_end_to_end_on_device(pkg, orc):
    """solveQP(Q) with both stages on the GPU: Phase-1 kernel, then the loop, nothing returns to the host in between"""
    cfg = pkg.CONFIGS["cfg4"]
    prob = pkg.generate_batch(cfg, 96)
    P, N, J = 96, cfg.N, cfg.J
    db = pkg.DeviceBatch(prob, np.zeros((P, N + J), dtype=np.int32), np.zeros((P, N)))
    st = db.phase1()
    db.solve()
    r = db.results()
    assert (st.cpu().numpy() == 1).all()
    x0, S0, _ = pkg.phase1_batch(prob)
    zo, So, sto, _, _ = oracle_batch(orc, prob, S0, x0)
    assert_parity(r["z"], r["S"], r["status"], zo, So, sto)


def test_host_buffer_chunked_and_resident_handle(pkg, orc):
    """ssqp_solve_batch_f64 uploads a large batch in chunks and solves them on internal launch lanes; the resident
    handle (ssqp_problem_upload / _solve / _set_vector) solves without moving V again: same results either way"""
    cfg = pkg.GenConfig(256, 1, 4, 512, 1e-3, 5.0 / 256, 1.05, 0.1)
    prob = pkg.generate_batch(cfg, 300, 8675309)            # 300 x 512 KiB = 150 MiB of V: more than one chunk
    x0, S0, st = pkg.phase1_batch(prob)
    assert (st == 1).all()
    z, S, status, detail = pkg.solveQP_batch(prob, S0, x0)
    zo, So, sto, _, _ = oracle_batch(orc, prob, S0, x0)
    assert_parity(z, S, status, zo, So, sto)
    rb = pkg.ResidentBatch(prob)
    z2, S2, st2, _ = rb.solve(S0, x0)
    assert np.array_equal(z2, z) and np.array_equal(S2, S) and np.array_equal(st2, status)
    z3, S3, st3, _ = rb.solve(S, z)                          # warm start at the optimum: a fixed point
    assert np.array_equal(S3, S) and (st3 >= 1).all() and (st3 <= 3).all()
    q2 = 2.0 * prob["q"]                                     # an efficient-frontier step: only q is replaced
    rb.set_vector("q", q2)
    z4, S4, st4, _ = rb.solve(S0, x0)
    prob2 = dict(prob)
    prob2["q"] = q2
    zo2, So2, sto2, _, _ = oracle_batch(orc, prob2, S0, x0)
    assert_parity(z4, S4, st4, zo2, So2, sto2)
    rb.close()


def test_lazy_handover_agrees_with_eager(pkg, orc):
    """"lazy_handover": the launches on the wavefront kernel's hand-over list (big-factor build, then workgroup kernel)
    are issued by ssqp_sync / the next call and only when the list is not empty.  cfg3 hands every QP over (K -> 229 >
    the first build's factor), cfg1 none: both must give the eager results, also with two contexts interleaved on
    two streams."""
    import torch
    for name, nprob in (("cfg3", 48), ("cfg1", 64)):
        cfg = pkg.CONFIGS[name]
        b0, prob, x0, S0 = pkg.DeviceBatch.generated(cfg, nprob, 4242)
        b0.solve()
        ref = b0.results()
        dev = b0.S.device
        lanes = []
        for _ in range(2):
            c = pkg.Context(dev.index)
            c.set_option("lazy_handover", 1)
            lanes.append((b0.twin(c), torch.cuda.Stream(dev)))
        for rep in range(3):
            for b, st in lanes:
                with torch.cuda.stream(st):
                    b.solve()
        for b, st in lanes:
            with torch.cuda.stream(st):
                r = b.results()
            assert np.array_equal(r["S"], ref["S"]) and np.array_equal(r["status"], ref["status"])
            assert np.array_equal(r["z"], ref["z"])
            assert ((r["stats"]["path"] & 64) != 0).all() == (name == "cfg3")    # (cfg3: every QP is handed over)
        full = dict(prob)
        full["V"] = b0.t["V"].cpu().numpy()
        zo, So, sto, _, _ = oracle_batch(orc, full, S0, x0)
        assert_parity(ref["z"], ref["S"], ref["status"], zo, So, sto)
