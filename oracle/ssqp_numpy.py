"""numpy / scipy-LAPACK restatement of solveQP(Q,S,x0) -- TEST INFRASTRUCTURE ONLY.

A second, independent restatement of the reference's hot loop
(src/SSQP.jl:237-377 and helpers :10-188, src/utils.jl:49-86) that calls the
SAME LAPACK routines Julia's LinearAlgebra calls (potrf/potri for
inv(cholesky(.)), BLAS gemm/gemv for `*`, pivoted-QR least squares for `\\`).
It exists to pin oracle/ssqp_oracle.c (plain C loops): both must take the same
status decisions and agree on z to rounding.  It is slow (Python loop per
iteration) and is used on small cases only.
"""
import numpy as np
from scipy.linalg import lapack, lstsq

IN, DN, UP, OE, EO = 0, 1, 2, 3, 4


def inv_cholesky(a):
    """inv(cholesky(A)): potrf('U') + potri('U') + mirror (LinearAlgebra.inv(::Cholesky))."""
    if a.shape[0] == 0:
        return a.copy()
    c, info = lapack.dpotrf(a, lower=0)
    if info != 0:
        raise np.linalg.LinAlgError("PosDefException(%d)" % info)
    ci, info = lapack.dpotri(c, lower=0)
    ci = np.triu(ci)
    return ci + np.triu(ci, 1).T


def getRowsGJr(X, tol=2.0 ** -33):
    """src/utils.jl:49-86"""
    A = np.array(X, dtype=np.float64, order="F", copy=True)
    nr, nc = A.shape
    rows = []
    c0 = list(range(nc))
    l1 = 0
    i = j = 0
    while i < nr and j < nc:
        v = np.abs(A[i, c0[j:]])
        mj = int(np.argmax(v))          # first maximum
        m = v[mj]
        mj += j
        if m <= tol:
            i += 1
        else:
            rows.append(i)
            c0[mj], c0[j] = c0[j], c0[mj]
            n = c0[j]
            cols = c0[j:]
            A[i, cols] = A[i, cols] / A[i, n]
            for k in range(nr):
                if k != i:
                    A[k, cols] = A[k, cols] - A[k, n] * A[i, cols]
            l1 = j + 1
            i += 1
            j += 1
    return rows, l1


def solveQP_warm(V, A, G, q, b, g, d, u, S, x0, maxIter=7777, tol=2.0 ** -26, tolG=2.0 ** -33):
    """Returns z, S, status, trace[(K, W, kind, id)] like oracle.solveQP_warm."""
    V = np.asarray(V, dtype=np.float64)
    N = V.shape[0]
    A = np.asarray(A, dtype=np.float64).reshape(-1, N)
    G = np.asarray(G, dtype=np.float64).reshape(-1, N)
    q, b, g, d, u = (np.asarray(x, dtype=np.float64).ravel() for x in (q, b, g, d, u))
    M, J = A.shape[0], G.shape[0]
    S = np.array(S, dtype=np.int32).copy()
    fu = u < np.inf
    fd = d > -np.inf
    z = np.array(x0, dtype=np.float64).copy()
    trace = []
    it = 0
    while True:
        it += 1
        if it > maxIter:
            return z, S, -it, trace
        Sz = S[:N]
        Se = S[N:]
        F = Sz == IN
        K = int(F.sum())
        if K == 0:                                             # freeK!  :35-59
            p = V @ z + q
            S0 = S.copy()
            t = True
            for k in range(N):
                if (p[k] >= -tol and S[k] == UP) or (p[k] <= tol and S[k] == DN):
                    S[k] = IN
                    t = False
            if t:
                trace.append((0, 0, 3, 0))
                return z, S, it, trace
            ipx = np.flatnonzero(S == IN)
            if len(ipx) > 0 and np.max(np.abs(p[ipx])) <= tol:
                S[ipx] = S0[ipx]
                trace.append((0, 0, 3, 0))
                return z, S, it, trace
            trace.append((0, 0, 0, 0))
            continue
        B = ~F
        Eg = Se == EO
        Og = Se == OE
        GE = G[Eg, :]
        AE = np.vstack([A[:, F], GE[:, F]])
        zB = z[B]
        AB = np.vstack([A[:, B], GE[:, B]])
        bE = np.concatenate([b, g[Eg]]) - AB @ zB
        ra, _la = getRowsGJr(np.hstack([AE, bE[:, None]]), tol)
        W0 = len(bE)
        W = len(ra)
        if W < W0:
            AE = AE[ra, :]
            bE = bE[ra]
            AB = AB[ra, :]
        try:
            iV = inv_cholesky(np.asfortranarray(V[np.ix_(F, F)]))
            VBF = V[np.ix_(B, F)]
            c = VBF.T @ zB + q[F]
            mT = iV @ AE.T
            Cm = AE @ mT
            Cm = (Cm + Cm.T) / 2
            Cm = inv_cholesky(np.asfortranarray(Cm))
        except np.linalg.LinAlgError:
            return z, S, -1, trace
        TC = mT @ Cm
        VQ = iV - mT @ TC.T
        alpha = TC @ bE - VQ @ c
        p = alpha - z[F]
        iF = np.flatnonzero(F)
        pn = np.max(np.abs(p)) if not np.isnan(p).any() else np.nan
        if pn > tolG:                                          # aStep!  :61-134
            Lo = []
            for k in range(K):
                j = iF[k]
                t = p[k]
                h = z[j]
                with np.errstate(all="ignore"):
                    dL = (d[j] - h) / t
                    uL = (u[j] - h) / t
                if t > tol and fu[j]:
                    Lo.append((uL, UP, j + 1))
                elif t < -tol and fd[j]:
                    Lo.append((dL, DN, j + 1))
            if J > 0:
                zo = g[Og] - G[Og, :] @ z
                po = G[np.ix_(Og, F)] @ p
                ik = np.flatnonzero(Og)
                for k in range(len(zo)):
                    if po[k] > tol:
                        Lo.append((zo[k] / po[k], EO, ik[k] + 1))
            L1 = 1.0
            if Lo:
                Lo.sort(key=lambda e: e[0])                    # stable
                L1 = Lo[0][0]
            if L1 < 1.0:
                z[F] += L1 * p
                fid = 0
                for (L, To, k) in Lo:
                    if L - L1 > tol:
                        break
                    if To == EO:
                        k += N
                    S[k - 1] = To
                    if k <= N:
                        z[k - 1] = d[k - 1] if To == DN else u[k - 1]
                    fid = k if fid == 0 else min(fid, k)
                trace.append((K, W, 1, fid))
                continue
            z[F] = alpha
        alphaL = -(TC.T @ c + Cm @ bE)
        gamma = VBF @ alpha + V[np.ix_(B, B)] @ zB + q[B] + AB.T @ alphaL
        iB = np.flatnonzero(B)                                 # KKTchk!  :136-188
        Li = []
        for k in range(len(gamma)):
            j = iB[k]
            t = gamma[k]
            if S[j] == UP and t > tolG:
                Li.append((-t, IN, j + 1))
            elif S[j] == DN and t < -tolG:
                Li.append((t, IN, j + 1))
        JE = GE.shape[0]
        if JE > 0:
            iE = [-1] * JE
            for pos, r in enumerate(ra):
                if r >= M:
                    iE[r - M] = pos
            ibx = np.flatnonzero(Eg)
            for j in range(JE):
                if iE[j] < 0:
                    x = lstsq(AE.T, GE[j, F], lapack_driver="gelsy")[0]
                    Lda = float(alphaL @ x)
                else:
                    Lda = alphaL[iE[j]]
                if Lda < -tolG:
                    Li.append((Lda, OE, ibx[j] + 1))
        if Li:
            Li.sort(key=lambda e: e[0])
            L, To, k = Li[0]
            if To == OE:
                k += N
            S[k - 1] = To
            trace.append((K, W, 2, k))
            continue
        for k in range(N):                                     # polishSz!  :10-32
            if S[k] == DN:
                z[k] = d[k]
            elif S[k] == UP:
                z[k] = u[k]
            else:
                if abs(z[k] - d[k]) < tol:
                    z[k] = d[k]
                    S[k] = DN
                elif abs(z[k] - u[k]) < tol:
                    z[k] = u[k]
                    S[k] = UP
        for j in range(J):
            S[N + j] = EO if abs(g[j] - z @ G[j, :]) < tol else OE
        trace.append((K, W, 3, 0))
        return z, S, it, trace
