#!/usr/bin/env python3
"""Generates tests/golden/*.npz (inputs + expected outputs) -- run in the build container only.

The reference is Julia and cannot be executed here (no Julia in the image), so the expected outputs are
NOT outputs of the reference.  They are outputs of oracle/ssqp_oracle.c that were, at generation time,
(1) reproduced decision-for-decision by the independent numpy/scipy-LAPACK restatement oracle/ssqp_numpy.py
    (same S, same iteration count, same per-iteration trace, z to 1e-11), and
(2) accepted by the independent KKT verifier tests/kkt.py (for the cases that converge).
The one result the reference itself pins (test/runtests.jl:23-32, S == [UP, IN, IN]) is case `kat3`.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
from kkt import assert_kkt  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle import ssqp_numpy as onp  # noqa: E402

pkg = ge.load_package()


def gen(N, M, J, T, ub, gscale, qscale, seed, delta=1e-3):
    cfg = pkg.GenConfig(N, M, J, T, delta, ub, gscale, qscale)
    p = pkg.generate_batch(cfg, 1, seed)
    return dict(V=p["V"][0], A=p["A"][0].reshape(N, M).T.copy(), G=p["G"][0].reshape(N, J).T.copy(), q=p["q"][0],
                b=p["b"][0], g=p["g"][0], d=p["d"][0], u=p["u"][0])


def cases():
    inf = np.inf
    c = {}
    V = np.array([[1 / 100, 1 / 80, 1 / 100], [1 / 80, 1 / 16, 1 / 40], [1 / 100, 1 / 40, 1 / 25]])
    c["kat3"] = dict(V=V, A=np.ones((1, 3)), G=np.zeros((0, 3)), q=np.zeros(3), b=np.ones(1), g=np.zeros(0),
                     d=np.zeros(3), u=np.array([0.7, inf, 0.7]))
    c["simplex12"] = gen(12, 1, 0, 24, 0.0, 1.2, 0.0, 11)
    c["box20_j3"] = gen(20, 1, 3, 40, 0.2, 1.02, 0.1, 12)
    c["eq2_j4_n30"] = gen(30, 2, 4, 60, 0.15, 1.0, 0.1, 13)
    c["tight_ineq_n24"] = gen(24, 1, 5, 48, 0.25, 0.93, 0.2, 14)
    c["box48_j6"] = gen(48, 1, 6, 96, 0.08, 0.97, 0.15, 15)
    # degenerate: two identical inequality rows (the rank filter purges one; multiplier via least squares)
    p = gen(16, 1, 3, 32, 0.3, 0.9, 0.2, 16)
    p["G"][1] = p["G"][0]
    p["g"][1] = p["g"][0]
    c["dup_rows_n16"] = p
    # K == 0 start (freeK! path): no equality, box only, start at the all-DN vertex
    p = gen(14, 0, 0, 28, 1.0, 1.0, 1.0, 17)
    p["q"] = p["q"] + 0.1 * (np.arange(14) % 3 == 0)      # mixed signs
    c["freek_n14"] = p
    # infeasible: sum z = 1 cannot be met with u = 0.01
    c["infeasible_n20"] = gen(20, 1, 0, 40, 0.01, 1.0, 0.0, 18)
    # free variables (d = -Inf, u = +Inf) next to bounded ones
    p = gen(10, 1, 2, 20, 0.0, 1.1, 0.3, 19)
    p["d"] = p["d"].copy()
    p["d"][:3] = -inf
    c["free_vars_n10"] = p
    return c


def main():
    for name, p in cases().items():
        x0, S0, st1 = orc.initQP(p["A"], p["G"], p["b"], p["g"], p["d"], p["u"])
        out = dict(p, x0=x0, S0=S0, phase1_status=np.int32(st1))
        if st1 == 1:
            z, S, status, det, tr = orc.solveQP_warm(p["V"], p["A"], p["G"], p["q"], p["b"], p["g"], p["d"], p["u"],
                                                    S0, x0, max_trace=4096)
            z2, S2, st2, tr2 = onp.solveQP_warm(p["V"], p["A"], p["G"], p["q"], p["b"], p["g"], p["d"], p["u"], S0, x0)
            assert status == st2 and np.array_equal(S, S2), (name, status, st2)
            assert [tuple(t) for t in tr] == [tuple(int(v) for v in t) for t in tr2], name
            assert np.abs(z - z2).max() <= 1e-11 * max(1.0, np.abs(z).max()), name
            assert status > 0, (name, status)
            assert_kkt(p["V"], p["A"], p["G"], p["q"], p["b"], p["g"], p["d"], p["u"], z, S)
            out.update(z=z, S=S, status=np.int64(status), trace=np.array(tr, dtype=np.int32).reshape(-1, 4))
            print("%-16s N=%3d iters=%3d K_final=%d purged=%s" % (
                name, len(p["q"]), status, tr[-1][0], any(t[1] < (p["A"].shape[0] + int((S0[len(p['q']):] == 4).sum())) for t in tr)))
        else:
            print("%-16s phase-1 status %d" % (name, st1))
        np.savez(os.path.join(HERE, name + ".npz"), **out)


if __name__ == "__main__":
    main()
