"""solveQP and batch drivers: the reference's entry points over the C ABI.

    solveQP(Q; settings, settingsLP)        src/SSQP.jl:224-234
    solveQP(Q, S, x0; settings)             src/SSQP.jl:237-377   <- the GPU hot path

Every solve goes through libssqp_hip.so; nothing here computes a solution on
the CPU.  PyTorch is used only to hold device memory / streams for the
device-resident batch entry point.
"""
import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import SSQPError
from .types import QP, Settings, Status

_default_ctx = None


def _csettings(s):
    s = s or Settings()
    if s.rule != "Dantzig":
        raise _capi.SSQPError("only rule=:Dantzig is implemented for Phase-1 (got :%s)" % s.rule)
    return _capi.CSettings(s.maxIter, 0, s.tol, s.tolG)


class Context:
    """One per GPU (ssqp_ctx).  Raises NoDeviceError when no GPU is present."""

    def __init__(self, device=None):
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self._h = C.c_void_p()
        _capi.check(_capi.lib().ssqp_ctx_create(device, C.byref(self._h)))

    def close(self):
        if self._h:
            _capi.lib().ssqp_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def set_option(self, name, value):
        """Per-context switch (ssqp_ctx_set_option): wave_kernel, wave_qp_per_cu, incremental, dense_gamma, wg_per_cu."""
        _capi.check(_capi.lib().ssqp_ctx_set_option(self._h, name.encode(), int(value)), self._h)

    def get_option(self, name):
        v = C.c_int(0)
        _capi.check(_capi.lib().ssqp_ctx_get_option(self._h, name.encode(), C.byref(v)), self._h)
        return v.value

    def options(self, **kw):
        """Context manager: set options, restore the previous values on exit."""
        ctx = self

        class _Scope:
            def __enter__(self):
                self.old = {k: ctx.get_option(k) for k in kw}
                for k, v in kw.items():
                    ctx.set_option(k, v)
                return ctx

            def __exit__(self, *exc):
                for k, v in self.old.items():
                    ctx.set_option(k, v)
                return False
        return _Scope()

    def last_kernel_ms(self):
        ms = C.c_float(0)
        _capi.check(_capi.lib().ssqp_last_kernel_ms(self._h, C.byref(ms)), self._h)
        return ms.value

    def recent_kernel_ms(self, n):
        """durations (ms) of the solve kernels of the last n launches on this context, oldest first (n <= 16)"""
        out = []
        for back in range(n - 1, -1, -1):
            ms = C.c_float(0)
            _capi.check(_capi.lib().ssqp_recent_kernel_ms(self._h, back, C.byref(ms)), self._h)
            out.append(ms.value)
        return out

    def flush(self):
        """lazy_handover: issue what the last call still owes before its in/out buffers are reused"""
        _capi.check(_capi.lib().ssqp_flush(self._h), self._h)

    def flush_to(self, stream):
        """flush(), and `stream` (a raw hipStream_t) then waits for the owed launch if that went out on another stream"""
        _capi.check(_capi.lib().ssqp_flush_to(self._h, C.c_void_p(stream)), self._h)

    def sync(self, stream=None):
        _capi.check(_capi.lib().ssqp_sync(self._h, stream), self._h)


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def solveQP(Q, S=None, x0=None, settings=None, settingsLP=None, ctx=None, return_detail=False):
    """z, S, status = solveQP(Q)            Phase-1 on the host, loop on the GPU
       z, S, status = solveQP(Q, S, x0)     warm start: the hot path only; S is mutated in place

    status > 0: iterations used; 0 infeasible; -1 numerical/model error; -(maxIter+1) iteration limit.
    """
    if not isinstance(Q, QP):
        raise TypeError("solveQP expects a QP")
    ctx = ctx or default_context()
    lib = _capi.lib()
    cs = _csettings(settings)
    N, M, J = Q.N, Q.M, Q.J
    z = np.zeros(N)
    status = C.c_int64(0)
    detail = C.c_int32(0)
    if (S is None) != (x0 is None):
        raise TypeError("solveQP(Q, S, x0): S and x0 go together")
    if S is None:
        csl = _csettings(settingsLP or settings)
        if Q.mc <= 0:  # the reference returns fill(DN, N) here (SSQP.jl:227)
            Sout = np.full(N, int(Status.DN), dtype=np.int32)
        else:
            Sout = np.zeros(N + J, dtype=np.int32)
        rc = lib.ssqp_solve_full_f64(ctx.handle, N, M, J, _p(Q.V), _p(Q.A), _p(Q.G), _p(Q.q), _p(Q.b), _p(Q.g),
                                     _p(Q.d), _p(Q.u), Q.mc, _p(Sout), _p(z), C.byref(cs), C.byref(csl),
                                     C.byref(status), C.byref(detail))
        _capi.check(rc, ctx.handle)
    else:
        if not (isinstance(S, np.ndarray) and S.dtype == np.int32 and S.flags.c_contiguous and S.size == N + J):
            raise TypeError("S must be a contiguous int32 array of length N+J (Vector{Status})")
        Sout = S
        x0 = _f64(x0)
        rc = lib.ssqp_solve_f64(ctx.handle, N, M, J, _p(Q.V), _p(Q.A), _p(Q.G), _p(Q.q), _p(Q.b), _p(Q.g), _p(Q.d),
                                _p(Q.u), _p(Sout), _p(x0), _p(z), C.byref(cs), C.byref(status), C.byref(detail))
        _capi.check(rc, ctx.handle)
    if return_detail:
        return z, Sout, int(status.value), int(detail.value)
    return z, Sout, int(status.value)


# ----------------------------------------------------------------------------
# batches of equal-shape QPs.  Arrays are "column-major blocks": V[p] is the
# N x N column-major image (symmetric, so V[p] == V[p].T), A[p] has shape (N, M)
# with A[p][j, r] = A_p[r, j], G[p] likewise (N, J).
# ----------------------------------------------------------------------------
class GenConfig:
    """Synthetic problem family; see ssqp_generate_problem."""

    def __init__(self, N, M=1, J=0, T=None, delta=1e-3, ub=0.0, gscale=1.2, qscale=0.0):
        self.N, self.M, self.J = int(N), int(M), int(J)
        self.T = int(T if T is not None else 2 * N)
        self.delta, self.ub, self.gscale, self.qscale = float(delta), float(ub), float(gscale), float(qscale)

    def c(self):
        return _capi.CGenCfg(self.N, self.M, self.J, self.T, self.delta, self.ub, self.gscale, self.qscale)


BASE_SEED = 20261003

CONFIGS = {
    "cfg1": GenConfig(50, 1, 0, 100, 1e-3, 0.0, 1.2, 0.0),
    "cfg2": GenConfig(512, 1, 10, 1024, 1e-3, 3 / 64, 1.2, 0.1),
    "cfg3": GenConfig(256, 1, 0, 512, 1e-3, 3 / 32, 1.2, 0.0),
    "cfg4": GenConfig(512, 1, 10, 1024, 1e-3, 3 / 64, 1.2, 0.1),
    "cfg4_j0": GenConfig(512, 1, 0, 1024, 1e-3, 3 / 64, 1.2, 0.1),
    "cfg5": GenConfig(2048, 8, 64, 1024, 0.0, 3 / 128, 1.2, 0.1),
}


def generate_batch(cfg, nprob, seed0=BASE_SEED, nthreads=0, with_V=True):
    """dict of numpy arrays V,A,G,q,b,g,d,u for problems seed0 .. seed0+nprob-1.  with_V=False leaves V out
    (it is the expensive part; DeviceBatch.generated makes it on the GPU, bit-identical)."""
    N, M, J = cfg.N, cfg.M, cfg.J
    out = dict(V=np.zeros((nprob, N, N)) if with_V else np.zeros((0, N, N)), A=np.zeros((nprob, N, M)),
               G=np.zeros((nprob, N, J)),
               q=np.zeros((nprob, N)), b=np.zeros((nprob, M)), g=np.zeros((nprob, J)), d=np.zeros((nprob, N)),
               u=np.zeros((nprob, N)))
    cc = cfg.c()
    rc = _capi.lib().ssqp_generate_batch(C.byref(cc), seed0, nprob, *[_p(out[k]) for k in "VAGqbgdu"], nthreads)
    _capi.check(rc)
    return out


def phase1_batch(prob, settingsLP=None, nthreads=0):
    """initQP for every problem of a batch (host C++): x0 (P,N), S (P,N+J) int32, status (P,)."""
    P, N = prob["q"].shape
    M, J = prob["b"].shape[1], prob["g"].shape[1]
    x0 = np.zeros((P, N))
    S = np.zeros((P, N + J), dtype=np.int32)
    st = np.zeros(P, dtype=np.int32)
    cs = _csettings(settingsLP)
    rc = _capi.lib().ssqp_phase1_batch_f64(P, N, M, J, *[_p(_f64(prob[k])) for k in "AGbgdu"], C.byref(cs), _p(x0),
                                           _p(S), _p(st), nthreads)
    _capi.check(rc)
    return x0, S, st


def solveQP_batch(prob, S, x0, settings=None, ctx=None, want_stats=False, want_mult=False):
    """Hot path on a batch held in host memory.  Returns z (P,N), S (P,N+J) copy, status (P,), detail (P,)[, stats]
    [, lambda (P,M+J), gamma (P,N): the multipliers of the last pass by row / variable id, include/ssqp_hip.h]."""
    ctx = ctx or default_context()
    P, N = prob["q"].shape
    M, J = prob["b"].shape[1], prob["g"].shape[1]
    arrs = [_f64(prob[k]) for k in "VAGqbgdu"]
    S = np.ascontiguousarray(S, dtype=np.int32).copy()
    x0 = _f64(x0)
    z = np.zeros((P, N))
    status = np.zeros(P, dtype=np.int64)
    detail = np.zeros(P, dtype=np.int32)
    stats = (_capi.CStats * P)()
    cs = _csettings(settings)
    lam = np.zeros((P, M + J)) if want_mult else None
    gam = np.zeros((P, N)) if want_mult else None
    rc = _capi.lib().ssqp_solve_batch_f64(ctx.handle, P, N, M, J, *[_p(a) for a in arrs], _p(S), _p(x0), _p(z),
                                          C.byref(cs), _p(status), _p(detail), C.cast(stats, C.c_void_p),
                                          lam.ctypes.data_as(C.c_void_p) if want_mult else None,
                                          gam.ctypes.data_as(C.c_void_p) if want_mult else None)
    _capi.check(rc, ctx.handle)
    out = (z, S, status, detail)
    if want_stats:
        out += (stats_to_numpy(stats),)
    if want_mult:
        out += (lam, gam)
    return out


class ResidentBatch:
    """A batch kept in HBM behind the C ABI (ssqp_problem_upload): solve it repeatedly from host (S, x0) without
    moving V again -- the warm-start entry solveQP(Q, S, x0) (SSQP.jl:237) for a whole batch."""
    _VEC = dict(q=0, b=1, g=2, d=3, u=4)

    def __init__(self, prob, ctx=None):
        self.ctx = ctx or default_context()
        self.P, self.N = prob["q"].shape
        self.M, self.J = prob["b"].shape[1], prob["g"].shape[1]
        arrs = [_f64(prob[k]) for k in "VAGqbgdu"]
        self._h = C.c_void_p()
        _capi.check(_capi.lib().ssqp_problem_upload(self.ctx.handle, self.P, self.N, self.M, self.J,
                                                    *[_p(a) for a in arrs], C.byref(self._h)), self.ctx.handle)

    def set_vector(self, name, data):
        data = _f64(data)
        _capi.check(_capi.lib().ssqp_problem_set_vector(self._h, self._VEC[name], _p(data)), self.ctx.handle)

    def solve(self, S, x0, settings=None):
        S = np.ascontiguousarray(S, dtype=np.int32).copy()
        x0 = _f64(x0)
        z = np.zeros((self.P, self.N))
        status = np.zeros(self.P, dtype=np.int64)
        detail = np.zeros(self.P, dtype=np.int32)
        cs = _csettings(settings)
        _capi.check(_capi.lib().ssqp_problem_solve(self._h, _p(S), _p(x0), _p(z), C.byref(cs), _p(status), _p(detail),
                                                   None), self.ctx.handle)
        return z, S, status, detail

    def close(self):
        if self._h:
            _capi.lib().ssqp_problem_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solveQP_batch_multi(prob, S, x0, ctxs, settings=None):
    """The batch cut into contiguous blocks over several contexts (one per GPU; ssqp_solve_batch_multi_f64)."""
    P, N = prob["q"].shape
    M, J = prob["b"].shape[1], prob["g"].shape[1]
    arrs = [_f64(prob[k]) for k in "VAGqbgdu"]
    S = np.ascontiguousarray(S, dtype=np.int32).copy()
    x0 = _f64(x0)
    z = np.zeros((P, N))
    status = np.zeros(P, dtype=np.int64)
    detail = np.zeros(P, dtype=np.int32)
    cs = _csettings(settings)
    hs = (C.c_void_p * len(ctxs))(*[c.handle for c in ctxs])
    rc = _capi.lib().ssqp_solve_batch_multi_f64(hs, len(ctxs), P, N, M, J, *[_p(a) for a in arrs], _p(S), _p(x0),
                                                _p(z), C.byref(cs), _p(status), _p(detail), None)
    _capi.check(rc, ctxs[0].handle)
    return z, S, status, detail


STATS_DTYPE = np.dtype([("iters", "<i8"), ("alg_bytes", "<i8"), ("read_bytes", "<i8"), ("alg_flops", "<i8"),
                        ("sum_k3", "<i8"),
                        ("max_k", "<i4"), ("path", "<i4")])
TRACE_DTYPE = np.dtype([("K", "<i4"), ("W", "<i4"), ("kind", "<i4"), ("id", "<i4")])


def stats_to_numpy(stats):
    return np.frombuffer(bytes(stats), dtype=STATS_DTYPE).copy()


class DeviceBatch:
    """A batch resident in HBM (torch tensors hold the memory) + the launch of the in-kernel loop."""

    @classmethod
    def generated(cls, cfg, nprob, seed0=BASE_SEED, ctx=None, ntrace=0, device=None, nthreads=0):
        """Synthetic batch with V generated ON the GPU (same bits as the host generator); the small arrays
        and the Phase-1 vertex come from the host.  Returns (batch, prob_without_V, x0, S0)."""
        prob = generate_batch(cfg, nprob, seed0, nthreads=nthreads, with_V=False)
        x0, S0, st = phase1_batch(prob, nthreads=nthreads)
        if not (st == 1).all():
            raise SSQPError("Phase-1 failed on a synthetic problem")
        self = cls(prob, S0, x0, ctx=ctx, ntrace=ntrace, device=device, _gen=(cfg, seed0))
        return self, prob, x0, S0

    def __init__(self, prob, S0, x0, ctx=None, ntrace=0, device=None, _gen=None):
        import torch
        self.torch = torch
        self.ctx = ctx or default_context()
        dev = torch.device("cuda", self.ctx.device if device is None else device)
        self.P, self.N = np.shape(x0)
        self.M, self.J = prob["b"].shape[1], prob["g"].shape[1]
        self.t = {k: torch.from_numpy(_f64(prob[k])).to(dev) for k in "AGqbgdu"}
        if _gen is None:
            self.t["V"] = torch.from_numpy(_f64(prob["V"])).to(dev)
        else:
            cfg, seed0 = _gen
            self.t["V"] = torch.empty((self.P, self.N, self.N), dtype=torch.float64, device=dev)
            cc = cfg.c()
            rc = _capi.lib().ssqp_generate_V_dev(self.ctx.handle, C.byref(cc), seed0, self.P,
                                                 C.c_void_p(self.t["V"].data_ptr()),
                                                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _capi.check(rc, self.ctx.handle)
        self.S0 = torch.from_numpy(np.ascontiguousarray(S0, dtype=np.int32)).to(dev)
        self.x0 = torch.from_numpy(_f64(x0)).to(dev)
        self._alloc_outputs(dev)
        self.detail = torch.zeros(self.P, dtype=torch.int32, device=dev)
        self.stats = torch.zeros((self.P, STATS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        self.ntrace = int(ntrace)
        self.trace = torch.zeros((self.P, max(self.ntrace, 1), 4), dtype=torch.int32, device=dev)
        self.lam = self.gam = None   # multiplier outputs (want_multipliers())
        self.use_stats = True        # False: the launch asks for no statistics (and, with ntrace = 0, gets the lean kernel builds)
        torch.cuda.synchronize(dev)

    def want_multipliers(self):
        """Allocate the optional multiplier outputs: lambda (P, M+J) and gamma (P, N), filled by every solve()."""
        torch = self.torch
        dev = self.S.device
        self.lam = torch.zeros((self.P, max(self.M + self.J, 1)), dtype=torch.float64, device=dev)
        self.gam = torch.zeros((self.P, self.N), dtype=torch.float64, device=dev)
        return self

    def _alloc_outputs(self, dev):
        """z, S, status are views of ONE byte buffer (`out`): the final gather of a sharded batch sends it as it is
        (dist.PackedGather: one collective, no packing copy)"""
        from . import dist as _dist
        torch = self.torch
        self.out = torch.zeros(_dist.packed_layout(self.P, self.N, self.J)[3], dtype=torch.uint8, device=dev)
        self.z, self.S, self.status = _dist.packed_views(self.out, self.P, self.N, self.J)

    def twin(self, ctx):
        """A second launch lane over the SAME inputs (V, A, G, ..., S0, x0 are shared, not copied) with its own
        context (workspace, work counter) and its own outputs: batches issued on different HIP streams can then
        overlap -- the drain of one launch (workgroups finish at different times) with the ramp-up of the next."""
        torch = self.torch
        o = object.__new__(DeviceBatch)
        o.torch, o.ctx = torch, ctx
        o.P, o.N, o.M, o.J = self.P, self.N, self.M, self.J
        o.t, o.S0, o.x0 = self.t, self.S0, self.x0
        o._alloc_outputs(self.S0.device)
        o.detail = torch.zeros_like(self.detail)
        o.stats = torch.zeros_like(self.stats)
        o.ntrace = self.ntrace
        o.trace = torch.zeros_like(self.trace)
        o.lam = o.gam = None
        o.use_stats = self.use_stats
        torch.cuda.synchronize(self.S0.device)
        return o

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr()) if t.numel() else None

    def solve(self, settings=None, stream=None):
        """One pass of the hot path over the batch (asynchronous on `stream`, default torch's current stream).
        Arrays given with a leading dimension of 1 are shared by every problem of the batch (stride 0)."""
        torch = self.torch
        cs = _csettings(settings)
        # S is in/out: the reset runs on the SAME stream as the launch (a raw hipStream_t is wrapped, so the copy
        # cannot race with the kernels of this or the previous launch on that stream)
        if stream is None:
            tstream = torch.cuda.current_stream(self.S.device)
        elif isinstance(stream, int):
            tstream = torch.cuda.ExternalStream(stream, device=self.S.device)
        else:
            tstream = stream
        # lazy hand-over: what the previous call on this context still owes goes out first, and this call's stream waits
        # for it even when that call ran on ANOTHER stream (the owed stages read and write S and z)
        self.ctx.flush_to(tstream.cuda_stream)
        with torch.cuda.stream(tstream):
            self.S.copy_(self.S0)
        stream = tstream.cuda_stream
        t = self.t
        per = dict(V=self.N * self.N, A=self.M * self.N, G=self.J * self.N, q=self.N, b=self.M, g=self.J, d=self.N,
                   u=self.N)
        strides = _capi.CStrides(*[(0 if (t[k].shape[0] == 1 and self.P > 1) else per[k]) for k in "VAGqbgdu"])
        rc = _capi.lib().ssqp_solve_batch_strided_dev_f64(
            self.ctx.handle, self.P, self.N, self.M, self.J, *[self._ptr(t[k]) for k in "VAGqbgdu"], C.byref(strides),
            self._ptr(self.S), self._ptr(self.x0), self._ptr(self.z), C.byref(cs), self._ptr(self.status),
            self._ptr(self.detail), self._ptr(self.stats) if self.use_stats else None, self._ptr(self.trace) if self.ntrace else None,
            self.ntrace, self._ptr(self.lam) if self.lam is not None else None,
            self._ptr(self.gam) if self.gam is not None else None, C.c_void_p(stream))
        _capi.check(rc, self.ctx.handle)

    def solve_full(self, settings=None, settingsLP=None, stream=None):
        """solveQP(Q) for the batch in ONE launch per QP (ssqp_solve_full_batch_dev_f64): Phase-1 and the loop in the same
        kernel; S, z, status are outputs (S0 / x0 are not read).  Asynchronous on `stream`."""
        torch = self.torch
        cs = _csettings(settings)
        csl = _csettings(settingsLP if settingsLP is not None else settings)
        if stream is None:
            stream = torch.cuda.current_stream(self.S.device).cuda_stream
        elif not isinstance(stream, int):
            stream = stream.cuda_stream
        self.ctx.flush_to(stream)
        t = self.t
        for k in "VAGqbgdu":
            if t[k].shape[0] == 1 and self.P > 1:
                raise SSQPError("DeviceBatch.solve_full needs per-problem arrays (no stride-0 sharing)")
        rc = _capi.lib().ssqp_solve_full_batch_dev_f64(
            self.ctx.handle, self.P, self.N, self.M, self.J, *[self._ptr(t[k]) for k in "VAGqbgdu"],
            self._ptr(self.S), self._ptr(self.z), C.byref(cs), C.byref(csl), self._ptr(self.status),
            self._ptr(self.detail), self._ptr(self.stats) if self.use_stats else None, self._ptr(self.lam) if self.lam is not None else None,
            self._ptr(self.gam) if self.gam is not None else None, C.c_void_p(stream))
        _capi.check(rc, self.ctx.handle)

    def phase1(self, settingsLP=None, stream=None):
        """initQP for the whole batch ON the GPU (ssqp_phase1_batch_dev_f64): fills self.x0 / self.S0 in place
        (asynchronous) and returns the status tensor (1 feasible, 0 infeasible, -1 singular basis)."""
        torch = self.torch
        cs = _csettings(settingsLP)
        if stream is None:
            stream = torch.cuda.current_stream(self.S.device).cuda_stream
        if not hasattr(self, "p1status"):
            self.p1status = torch.zeros(self.P, dtype=torch.int32, device=self.S.device)
        t = self.t
        for k in "AGbgdu":
            if t[k].shape[0] == 1 and self.P > 1:
                raise SSQPError("DeviceBatch.phase1 needs per-problem arrays (no stride-0 sharing)")
        rc = _capi.lib().ssqp_phase1_batch_dev_f64(
            self.ctx.handle, self.P, self.N, self.M, self.J, *[self._ptr(t[k]) for k in "AGbgdu"], C.byref(cs),
            self._ptr(self.x0), self._ptr(self.S0), self._ptr(self.p1status), C.c_void_p(stream))
        _capi.check(rc, self.ctx.handle)
        return self.p1status

    def results(self):
        self.ctx.sync(C.c_void_p(self.torch.cuda.current_stream(self.S.device).cuda_stream))   # (settles a lazy hand-over)
        self.torch.cuda.synchronize(self.S.device)
        stats = np.frombuffer(self.stats.cpu().numpy().tobytes(), dtype=STATS_DTYPE).copy()
        trace = self.trace.cpu().numpy() if self.ntrace else None
        out = dict(z=self.z.cpu().numpy(), S=self.S.cpu().numpy(), status=self.status.cpu().numpy(),
                   detail=self.detail.cpu().numpy(), stats=stats, trace=trace)
        if self.lam is not None:
            out["lam"] = self.lam.cpu().numpy()[:, :self.M + self.J]
            out["gam"] = self.gam.cpu().numpy()
        return out
