"""ctypes binding of oracle/libssqp_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module (see ssqp_oracle.c header).  The product never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libssqp_oracle.so")

IN, DN, UP, OE, EO = 0, 1, 2, 3, 4


class Settings(C.Structure):
    """src/types.jl:390-408 (Float64 defaults)."""
    _fields_ = [("maxIter", C.c_int32), ("rule", C.c_int32), ("tol", C.c_double), ("tolG", C.c_double)]

    def __init__(self, maxIter=7777, tol=2.0 ** -26, tolG=2.0 ** -33):
        super().__init__(maxIter, 0, tol, tolG)


class Trace(C.Structure):
    _fields_ = [("K", C.c_int32), ("W", C.c_int32), ("kind", C.c_int32), ("id", C.c_int32)]


def build(force=False):
    src = os.path.join(_HERE, "ssqp_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int32)
        _lib.orc_solveQP_warm.restype = C.c_int64
        _lib.orc_solveQP_warm.argtypes = [C.c_int] * 3 + [dp] * 8 + [ip, dp, dp, C.POINTER(Settings), ip,
                                                                 C.POINTER(Trace), C.c_int, C.POINTER(C.c_int)]
        _lib.orc_solveQP.restype = C.c_int64
        _lib.orc_solveQP.argtypes = [C.c_int] * 3 + [dp] * 8 + [C.c_int, ip, dp, C.POINTER(Settings), ip,
                                                            C.POINTER(Trace), C.c_int, C.POINTER(C.c_int)]
        _lib.orc_initQP.restype = C.c_int
        _lib.orc_initQP.argtypes = [C.c_int] * 3 + [dp] * 6 + [C.c_double, dp, ip]
        _lib.orc_getRowsGJr.restype = C.c_int
        _lib.orc_getRowsGJr.argtypes = [dp, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _lib.orc_solveQP_warm_batch.restype = C.c_int
        _lib.orc_solveQP_warm_batch.argtypes = [C.c_int] * 4 + [dp] * 8 + [ip, dp, dp, C.POINTER(Settings),
                                                                           C.POINTER(C.c_int64), ip, C.c_int]
        _lib.orc_solveQP_warm_batch2.restype = C.c_int
        _lib.orc_solveQP_warm_batch2.argtypes = [C.c_int] * 4 + [dp] * 8 + [ip, dp, dp, C.POINTER(Settings),
                                                                            C.POINTER(C.c_int64), ip, C.c_int, C.c_int]
        _lib.orc_solveQP_warm_batch3.restype = C.c_int
        _lib.orc_solveQP_warm_batch3.argtypes = [C.c_int] * 4 + [dp] * 8 + [ip, dp, dp, C.POINTER(Settings),
                                                                            C.POINTER(C.c_int64), ip, C.c_int, C.c_int, dp, dp]
        _lib.orc_lapack_load.restype = C.c_int
        _lib.orc_lapack_load.argtypes = [C.c_char_p]
        _lib.orc_initQP_batch.restype = C.c_int
        _lib.orc_initQP_batch.argtypes = [C.c_int] * 4 + [dp] * 6 + [C.c_double, dp, ip, ip, C.c_int]
    return _lib


def _f(a):
    """column-major float64 copy (Julia layout)."""
    return np.asfortranarray(np.array(a, dtype=np.float64))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _norm(V, A, G, q, b, g, d, u):
    V = _f(V)
    N = V.shape[0]
    A = _f(A).reshape(-1, N, order="F") if np.size(A) else np.zeros((0, N), order="F")
    G = _f(G).reshape(-1, N, order="F") if np.size(G) else np.zeros((0, N), order="F")
    q, b, g, d, u = (np.ascontiguousarray(np.array(x, dtype=np.float64).ravel()) for x in (q, b, g, d, u))
    return V, A, G, q, b, g, d, u, N, A.shape[0], G.shape[0]


def getRowsGJr(X, tol=2.0 ** -33):
    X = _f(X)
    nr, nc = X.shape
    rows = (C.c_int * (nr + 1))()
    l1 = C.c_int(0)
    n = lib().orc_getRowsGJr(_dp(X), nr, nc, tol, rows, C.byref(l1))
    return [rows[i] for i in range(n)], l1.value


def initQP(A, G, b, g, d, u, tol=2.0 ** -26):
    N = len(d)
    _, A, G, _, b, g, d, u, N, M, J = _norm(np.zeros((N, N)), A, G, np.zeros(N), b, g, d, u)
    x = np.zeros(N)
    S = np.zeros(N + J, dtype=np.int32)
    st = lib().orc_initQP(N, M, J, _dp(A), _dp(G), _dp(b), _dp(g), _dp(d), _dp(u), tol, _dp(x), _ip(S))
    return x, S, st


def solveQP_warm(V, A, G, q, b, g, d, u, S, x0, settings=None, max_trace=0):
    """solveQP(Q, S, x0) (src/SSQP.jl:237).  Returns z, S (copy), status, detail, trace."""
    V, A, G, q, b, g, d, u, N, M, J = _norm(V, A, G, q, b, g, d, u)
    settings = settings or Settings()
    S = np.array(S, dtype=np.int32).copy()
    x0 = np.ascontiguousarray(np.array(x0, dtype=np.float64))
    z = np.zeros(N)
    det = C.c_int32(0)
    tr = (Trace * max(max_trace, 1))()
    nt = C.c_int(0)
    st = lib().orc_solveQP_warm(N, M, J, _dp(V), _dp(A), _dp(G), _dp(q), _dp(b), _dp(g), _dp(d), _dp(u),
                                _ip(S), _dp(x0), _dp(z), C.byref(settings), C.byref(det), tr, max_trace,
                                C.byref(nt))
    trace = [(tr[i].K, tr[i].W, tr[i].kind, tr[i].id) for i in range(min(nt.value, max_trace))]
    return z, S, int(st), int(det.value), trace


def solveQP(V, A, G, q, b, g, d, u, mc=1, settings=None, max_trace=0):
    """solveQP(Q) (src/SSQP.jl:224): Phase-1 + loop."""
    V, A, G, q, b, g, d, u, N, M, J = _norm(V, A, G, q, b, g, d, u)
    settings = settings or Settings()
    S = np.zeros(N + J, dtype=np.int32)
    z = np.zeros(N)
    det = C.c_int32(0)
    tr = (Trace * max(max_trace, 1))()
    nt = C.c_int(0)
    st = lib().orc_solveQP(N, M, J, _dp(V), _dp(A), _dp(G), _dp(q), _dp(b), _dp(g), _dp(d), _dp(u), mc,
                           _ip(S), _dp(z), C.byref(settings), C.byref(det), tr, max_trace, C.byref(nt))
    trace = [(tr[i].K, tr[i].W, tr[i].kind, tr[i].id) for i in range(min(nt.value, max_trace))]
    return z, S, int(st), int(det.value), trace


_lapack_state = None


def lapack_available():
    """Bind dpotrf/dpotri/dgemm/dgemv of the OpenBLAS that scipy bundles (what Julia's LinearAlgebra calls at
    SSQP.jl:322-331,351-352) into the oracle; False when the library cannot be found."""
    global _lapack_state
    if _lapack_state is None:
        _lapack_state = False
        try:
            import glob
            import scipy
            cands = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*"))
            cands += glob.glob(os.path.join(os.path.dirname(scipy.__file__), ".libs", "libopenblas*"))
            for c in cands:
                if lib().orc_lapack_load(os.path.abspath(c).encode()) == 0:
                    _lapack_state = True
                    break
        except Exception:
            _lapack_state = False
    return _lapack_state


def solveQP_warm_batch(V, A, G, q, b, g, d, u, S, x0, settings=None, nthreads=0, lapack=False, want_mult=False):
    """Back-to-back batch: V (P,N,N) with each V[p] symmetric (so C order == column-major),
    A (P,N,M) = per-problem column-major M x N, G (P,N,J) likewise; vectors (P,len).
    want_mult: also return (lambda (P,M+J), gamma (P,N)), the multipliers of the last pass by row / variable id."""
    P, N = q.shape
    M = b.shape[1]
    J = g.shape[1]
    settings = settings or Settings()
    S = np.ascontiguousarray(S, dtype=np.int32).copy()
    z = np.zeros((P, N))
    status = np.zeros(P, dtype=np.int64)
    detail = np.zeros(P, dtype=np.int32)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (V, A, G, q, b, g, d, u)]
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    if lapack and not lapack_available():
        raise RuntimeError("no LAPACK library to bind (scipy's OpenBLAS not found)")
    lam = np.zeros((P, M + J)) if want_mult else None
    gam = np.zeros((P, N)) if want_mult else None
    used = lib().orc_solveQP_warm_batch3(P, N, M, J, *[_dp(a) for a in arrs], _ip(S), _dp(x0), _dp(z),
                                         C.byref(settings), status.ctypes.data_as(C.POINTER(C.c_int64)),
                                         _ip(detail), nthreads, 1 if lapack else 0,
                                         _dp(lam) if want_mult else None, _dp(gam) if want_mult else None)
    if want_mult:
        return z, S, status, detail, used, lam, gam
    return z, S, status, detail, used


def initQP_batch(A, G, b, g, d, u, tol=2.0 ** -26, nthreads=0):
    P, N = d.shape
    M = b.shape[1]
    J = g.shape[1]
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (A, G, b, g, d, u)]
    x = np.zeros((P, N))
    S = np.zeros((P, N + J), dtype=np.int32)
    st = np.zeros(P, dtype=np.int32)
    lib().orc_initQP_batch(P, N, M, J, *[_dp(a) for a in arrs], tol, _dp(x), _ip(S), _ip(st), nthreads)
    return x, S, st
