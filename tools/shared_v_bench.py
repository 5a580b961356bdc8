"""GPU box: the efficient-frontier batch mode: ONE V, A, G, b, g, d, u for the whole batch, a
different q per QP (QP(P, q, L) of src/types.jl:303-319 swept over L) through the strided entry with stride 0.  V then
stays L2-resident, so this is also what the loop costs when its multiplier pass does not have to come from HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
import torch
from oracle import oracle as orc

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = pkg.CONFIGS["cfg4"]
base = pkg.generate_batch(cfg, 1, 31337)
x0b, S0b, st = pkg.phase1_batch(base)          # Phase-1 does not look at q or V: one vertex for the whole sweep
assert (st == 1).all()
Ls = np.linspace(0.25, 4.0, P)
shared = dict(base)
shared["q"] = np.ascontiguousarray(Ls[:, None] * base["q"][0][None, :])
S0 = np.ascontiguousarray(np.repeat(S0b, P, axis=0))
x0 = np.ascontiguousarray(np.repeat(x0b, P, axis=0))
for qpc, nl in ((4, 1), (8, 3)):
    ctxs = [pkg.Context(0) for _ in range(nl)]
    for c in ctxs:
        c.set_option("wave_qp_per_cu", qpc)
        c.set_option("lazy_handover", 1 if nl > 1 else 0)
    b0 = pkg.DeviceBatch(shared, S0, x0, ctx=ctxs[0])
    lanes = [(b0, torch.cuda.current_stream())] + [(b0.twin(c), torch.cuda.Stream()) for c in ctxs[1:]]

    def run(n):
        for i in range(n):
            b, s = lanes[i % nl]
            with torch.cuda.stream(s):
                b.solve()
        for b, s in lanes:
            b.ctx.sync(s.cuda_stream)
        torch.cuda.synchronize()
    run(2 * nl)
    t = time.perf_counter(); run(6 * nl); dt = (time.perf_counter() - t) / (6 * nl)
    r = b0.results()
    it = r["status"]
    print("shared V, %d QPs, %d per CU, %d lane(s): %.3f ms per batch -> %.0f QPs/s; converged %d, passes mean %.1f, "
          "distinct final S %d, kernel ms %.3f" % (P, qpc, nl, dt * 1e3, P / dt, int((it > 0).sum()), it.mean(),
                                                    len(set(map(bytes, r["S"]))), b0.ctx.last_kernel_ms()), flush=True)
# parity of a sample against the oracle
idx = np.linspace(0, P - 1, 16).astype(int)
full = {k: np.ascontiguousarray(np.repeat(base[k], len(idx), axis=0)) for k in "VAGbgdu"}
full["q"] = np.ascontiguousarray(shared["q"][idx])
zo, So, sto, _, _ = orc.solveQP_warm_batch(full["V"], full["A"], full["G"], full["q"], full["b"], full["g"], full["d"],
                                          full["u"], S0[idx], x0[idx])
ok = np.array_equal(r["S"][idx], So) and np.array_equal(r["status"][idx], sto)
err = np.abs(r["z"][idx] - zo).max() / max(1e-300, np.abs(zo).max())
print("sample of 16 against the oracle: S and pass counts equal %s, max rel z err %.2e" % (ok, err))
