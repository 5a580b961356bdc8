"""Debug helper (GPU box): compare HIP vs oracle on a batch, print the first trace divergence.
usage: debug_parity.py <cfgname | gen:N,M,J,T,delta,ub,gscale,qscale> nprob [seedshift] [opt=value ...]"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
from oracle import oracle as orc
pkg = ge.load_package()
name, nprob = sys.argv[1], int(sys.argv[2])
rest = sys.argv[3:]
shift = int(rest[0]) if rest and "=" not in rest[0] else 0
opts = {a.split("=")[0]: int(a.split("=")[1]) for a in rest if "=" in a}
seed0 = pkg.BASE_SEED + shift
if name.startswith("gen:"):
    v = name[4:].split(",")
    cfg = pkg.GenConfig(int(v[0]), int(v[1]), int(v[2]), int(v[3]), *[float(x) for x in v[4:]])
else:
    cfg = pkg.CONFIGS[name]
prob = pkg.generate_batch(cfg, nprob, seed0)
x0, S0, st = pkg.phase1_batch(prob)
ctx = pkg.default_context()
for k, val in opts.items():
    ctx.set_option(k, val)
db = pkg.DeviceBatch(prob, S0, x0, ntrace=1024)
db.solve()
res = db.results()
t0 = time.time()
for _ in range(3):
    db.solve()
res = db.results()
ms = (time.time() - t0) / 3 * 1e3
zo, So, sto, deto, _ = orc.solveQP_warm_batch(prob["V"], prob["A"], prob["G"], prob["q"], prob["b"], prob["g"], prob["d"], prob["u"], S0, x0)
bad = np.flatnonzero((res["status"] != sto) | (res["S"] != So).any(axis=1))
scale = np.maximum(np.abs(zo).max(axis=1), 1e-300)
rel = (np.abs(res["z"] - zo).max(axis=1) / scale)
print(name, opts, "sum_k3 %.4g" % float(res["stats"]["sum_k3"].sum()), "alg_flops %.4g" % float(res["stats"]["alg_flops"].sum()), "nprob", nprob, "bad", len(bad), bad[:20], "paths", np.unique(res["stats"]["path"]), "maxK", res["stats"]["max_k"].max(),
      "max rel z err %.2e" % rel.max(), "kernel ms %.3f wall ms %.3f" % (ctx.last_kernel_ms(), ms), flush=True)
for p in bad[:3]:
    A = prob["A"][p].reshape(cfg.N, cfg.M).T; G = prob["G"][p].reshape(cfg.N, cfg.J).T
    z, S, stt, det, tr = orc.solveQP_warm(prob["V"][p], A, G, prob["q"][p], prob["b"][p], prob["g"][p], prob["d"][p], prob["u"][p], S0[p], x0[p], max_trace=1024)
    got = [tuple(int(v) for v in r) for r in res["trace"][p][:min(1024, max(stt, 1) + 5)]]
    print("problem", p, "oracle status", stt, "hip status", res["status"][p], "detail", res["detail"][p], "path", res["stats"]["path"][p])
    for i, (a, b) in enumerate(zip(got, tr)):
        if a != b:
            print("  first divergence at iter", i + 1, "hip", got[max(0, i - 3):i + 3], "oracle", tr[max(0, i - 3):i + 3])
            break
    else:
        print("  traces equal over", min(len(got), len(tr)), "oracle len", len(tr), "S diff at", np.flatnonzero(res["S"][p] != So[p])[:10],
              "rel z", rel[p])
