"""Independent KKT verifier (test infrastructure): does (z, S) solve
    min 1/2 z'Vz + q'z  s.t. Az=b, Gz<=g, d<=z<=u ?
Nothing here shares code with the oracle or the product: multipliers are recovered by a least-squares fit of
the stationarity condition on the free variables, then signs and complementarity are checked."""
import numpy as np

IN, DN, UP, OE, EO = 0, 1, 2, 3, 4


def kkt_report(V, A, G, q, b, g, d, u, z, S, eps=1e-7):
    V = np.asarray(V, float)
    N = V.shape[0]
    A = np.asarray(A, float).reshape(-1, N)
    G = np.asarray(G, float).reshape(-1, N)
    M, J = A.shape[0], G.shape[0]
    z = np.asarray(z, float)
    Sz, Se = np.asarray(S[:N]), np.asarray(S[N:N + J])
    rep = {}
    rep["eq"] = float(np.abs(A @ z - b).max()) if M else 0.0
    rep["ineq"] = float(np.maximum(G @ z - g, 0).max()) if J else 0.0
    rep["box"] = float(max(np.maximum(d - z, 0).max(), np.maximum(z - u, 0).max()))
    grad = V @ z + q
    F = Sz == IN
    E = np.flatnonzero(G @ z - g > -eps) if J else np.zeros(0, int)     # active by residual, not by label
    C = np.vstack([A, G[E]]) if (M + len(E)) else np.zeros((0, N))
    dn, up = Sz == DN, Sz == UP
    if C.shape[0] and F.any():
        lam = np.linalg.lstsq(C[:, F].T, -grad[F], rcond=None)[0]
    else:
        lam = np.zeros(C.shape[0])
    gamma = grad + C.T @ lam
    if C.shape[0] and (np.linalg.matrix_rank(C[:, F]) < C.shape[0] if F.any() else True):
        # degenerate vertex (more active rows than independent free columns): the multipliers are not unique and the
        # least-norm ones need not be the sign-feasible ones -- the point is optimal iff SOME multipliers satisfy every
        # condition, which is a linear feasibility problem
        bad = ((len(E) and lam[M:].min() <= -eps) or (dn.any() and gamma[dn].min() <= -eps) or
               (up.any() and gamma[up].max() >= eps))
        if bad:
            from scipy.optimize import linprog
            t = eps / 4
            rows, rhs = [], []
            if F.any():
                rows += [C[:, F].T, -C[:, F].T]
                rhs += [-grad[F] + t, grad[F] + t]
            if dn.any():
                rows.append(-C[:, dn].T)
                rhs.append(grad[dn] + t)
            if up.any():
                rows.append(C[:, up].T)
                rhs.append(-grad[up] + t)
            bounds = [(None, None)] * M + [(-t, None)] * len(E)
            res = linprog(np.zeros(C.shape[0]), A_ub=np.vstack(rows), b_ub=np.concatenate(rhs), bounds=bounds,
                          method="highs")
            if res.status == 0:
                lam = res.x
                gamma = grad + C.T @ lam
                rep["multipliers"] = "linear feasibility (degenerate vertex)"
    rep["stationarity_free"] = float(np.abs(gamma[F]).max()) if F.any() else 0.0
    rep["mu_min"] = float(lam[M:].min()) if len(E) else 0.0             # inequality multipliers must be >= 0
    rep["gamma_dn_min"] = float(gamma[dn].min()) if dn.any() else 0.0   # >= 0 at lower bounds
    rep["gamma_up_max"] = float(gamma[up].max()) if up.any() else 0.0   # <= 0 at upper bounds
    rep["at_bound"] = float(max(np.abs(z[dn] - np.asarray(d)[dn]).max() if dn.any() else 0.0,
                                np.abs(z[up] - np.asarray(u)[up]).max() if up.any() else 0.0))
    rep["labels_eo"] = bool(all((abs(g[j] - G[j] @ z) < 2.0 ** -26) == (Se[j] == EO) for j in range(J)))
    rep["objective"] = float(0.5 * z @ V @ z + q @ z)
    return rep


def assert_kkt(V, A, G, q, b, g, d, u, z, S, eps=1e-7):
    r = kkt_report(V, A, G, q, b, g, d, u, z, S, eps)
    assert r["eq"] < eps and r["ineq"] < eps and r["box"] < eps, r
    assert r["stationarity_free"] < eps, r
    assert r["mu_min"] > -eps and r["gamma_dn_min"] > -eps and r["gamma_up_max"] < eps, r
    assert r["at_bound"] == 0.0 and r["labels_eo"], r
    return r
