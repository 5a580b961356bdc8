"""Data model of the reference, mirrored for the Python host side.

Reference: src/types.jl -- Status (:17-23), QP (:214-301), Settings (:390-408).
Names, defaults, checks and model codes (`mc`) are the reference's.
"""
import enum
import warnings

import numpy as np


class Status(enum.IntEnum):
    """@enum Status (types.jl:17-23); Int32 codes are the ABI of `S`."""
    IN = 0  # within the lower and upper bound
    DN = 1  # down, lower bound
    UP = 2  # upper bound
    OE = 3  # original <=, not active
    EO = 4  # edge, <= as =, active


IN, DN, UP, OE, EO = Status.IN, Status.DN, Status.UP, Status.OE, Status.EO


class DimensionMismatch(ValueError):
    pass


class Settings:
    """Settings{Float64} (types.jl:390-408).  Unknown keywords are rejected,
    like the reference's keyword constructor does."""
    __slots__ = ("maxIter", "tol", "tolG", "pivot", "rule")

    def __init__(self, maxIter=7777, tol=2.0 ** -26, tolG=2.0 ** -33, pivot="column", rule="Dantzig"):
        self.maxIter = int(maxIter)
        self.tol = float(tol)
        self.tolG = float(tolG)
        self.pivot = str(pivot)
        self.rule = str(rule)

    def __repr__(self):
        return "Settings(maxIter=%d, tol=%g, tolG=%g, pivot=:%s, rule=:%s)" % (
            self.maxIter, self.tol, self.tolG, self.pivot, self.rule)


class QP:
    """QP(V; q, u, d, G, g, A, b)  (types.jl:229-301)

        min (1/2) z'Vz + q'z   s.t.  Az = b,  Gz <= g,  d <= z <= u

    Defaults as in the reference: q = 0, u = +Inf, d = 0, no inequalities,
    A = ones(1,N), b = [1].  `mc` is the model code: 1 OK, -20 no bounds and no
    inequalities, -30 some d == u, -70 V not positive semidefinite.
    Like the reference, bounds with u < d are swapped IN THE CALLER'S arrays.
    """

    def __init__(self, V, q=None, u=None, d=None, G=None, g=None, A=None, b=None, check_psd=True, _mc=None):
        V = np.asarray(V, dtype=np.float64)
        N = V.shape[0]
        if V.ndim != 2 or V.shape != (N, N):
            raise DimensionMismatch("incompatible dimension: V")
        q = np.zeros(N) if q is None else q
        u = np.full(N, np.inf) if u is None else u
        d = np.zeros(N) if d is None else d
        G = np.ones((0, N)) if G is None else np.asarray(G, dtype=np.float64)
        g = np.ones(0) if g is None else g
        A = np.ones((1, N)) if A is None else np.asarray(A, dtype=np.float64)
        b = np.ones(1) if b is None else b
        M = int(np.size(b))
        J = int(np.size(g))
        mc = 1
        V = (V + V.T) / 2                                    # types.jl:243
        if _mc is None and check_psd and N > 0:
            if np.linalg.eigvalsh(V)[0] < 0:                 # types.jl:246-249 (strict, no tolerance)
                mc = -70
                warnings.warn("variance matrix is not positive-semidefinite")
        A = A.reshape(M, N) if A.size == M * N and A.ndim != 2 else A
        G = G.reshape(J, N) if G.size == J * N and G.ndim != 2 else G
        if A.shape != (M, N):
            raise DimensionMismatch("incompatible dimension: A")
        if G.shape != (J, N):
            raise DimensionMismatch("incompatible dimension: G")
        for name, v in (("q", q), ("d", d), ("u", u)):
            if np.shape(v)[0] != N:
                raise DimensionMismatch("incompatible dimension: " + name)
        da, ua = np.asarray(d, dtype=np.float64), np.asarray(u, dtype=np.float64)
        if np.any(da == ua):                                 # types.jl:275-278
            mc = -30
            warnings.warn("downside bound == upper bound detected")
        if not (J > 0 or np.any(np.isfinite(da)) or np.any(np.isfinite(ua))):   # types.jl:281-284
            mc = -20
            warnings.warn("no inequalities and bounds")
        iu = ua < da                                         # types.jl:286-292
        if iu.any():
            warnings.warn("swap the elements where u < d, to make sure u > d")
            if isinstance(u, np.ndarray) and isinstance(d, np.ndarray):
                t = u[iu].copy()
                u[iu] = d[iu]
                d[iu] = t
                da, ua = np.asarray(d, dtype=np.float64), np.asarray(u, dtype=np.float64)
            else:
                da, ua = np.where(iu, ua, da), np.where(iu, da, ua)
        self.V = np.asfortranarray(V)
        self.A = np.asfortranarray(np.array(A, dtype=np.float64))
        self.G = np.asfortranarray(np.array(G, dtype=np.float64))
        self.q = np.array(q, dtype=np.float64).ravel().copy()
        self.b = np.array(b, dtype=np.float64).ravel().copy()
        self.g = np.array(g, dtype=np.float64).ravel().copy()
        self.d = np.array(da, dtype=np.float64).ravel().copy()
        self.u = np.array(ua, dtype=np.float64).ravel().copy()
        self.N, self.M, self.J = N, M, J
        self.mc = mc if _mc is None else int(_mc)

    @classmethod
    def inner(cls, V, A, G, q, b, g, d, u, mc=1):
        """The 12-field inner constructor QP{T}(V,A,G,q,b,g,d,u,N,M,J,mc) (types.jl:214-227):
        no symmetrisation check, no eigmin test (examples/rwMOI.jl:323 uses it the same way)."""
        return cls(V, q=q, u=u, d=d, G=G, g=g, A=A, b=b, check_psd=False, _mc=mc)

    def __repr__(self):
        return "QP(N=%d, M=%d, J=%d, mc=%d)" % (self.N, self.M, self.J, self.mc)
