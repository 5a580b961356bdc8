"""Debug helper (GPU box): compare HIP vs oracle on a batch, print the first trace divergence."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
from oracle import oracle as orc
pkg = ge.load_package()
name, nprob = sys.argv[1], int(sys.argv[2])
seed0 = pkg.BASE_SEED + (int(sys.argv[3]) if len(sys.argv) > 3 else 0)
cfg = pkg.CONFIGS[name]
prob = pkg.generate_batch(cfg, nprob, seed0)
x0, S0, st = pkg.phase1_batch(prob)
db = pkg.DeviceBatch(prob, S0, x0, ntrace=1024)
db.solve()
res = db.results()
zo, So, sto, deto, _ = orc.solveQP_warm_batch(prob["V"], prob["A"], prob["G"], prob["q"], prob["b"], prob["g"], prob["d"], prob["u"], S0, x0)
bad = np.flatnonzero((res["status"] != sto) | (res["S"] != So).any(axis=1))
print("nprob", nprob, "bad", len(bad), bad[:20], "paths", np.unique(res["stats"]["path"]), "maxK", res["stats"]["max_k"].max())
for p in bad[:3]:
    A = prob["A"][p].reshape(cfg.N, cfg.M).T; G = prob["G"][p].reshape(cfg.N, cfg.J).T
    z, S, stt, det, tr = orc.solveQP_warm(prob["V"][p], A, G, prob["q"][p], prob["b"][p], prob["g"][p], prob["d"][p], prob["u"][p], S0[p], x0[p], max_trace=1024)
    got = [tuple(int(v) for v in r) for r in res["trace"][p][:min(1024, max(stt, 1) + 5)]]
    print("problem", p, "oracle status", stt, "hip status", res["status"][p], "detail", res["detail"][p])
    for i, (a, b) in enumerate(zip(got, tr)):
        if a != b:
            print("  first divergence at iter", i + 1, "hip", got[max(0, i - 2):i + 3], "oracle", tr[max(0, i - 2):i + 3])
            break
    else:
        print("  traces equal over", min(len(got), len(tr)), "oracle len", len(tr))
