"""GPU box: Phase-1 kernel vs host C++ (bit-identical?) and its throughput.  usage: dbg_phase1.py cfg nprob"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
import torch
name, nprob = sys.argv[1], int(sys.argv[2])
cfg = pkg.CONFIGS[name]
prob = pkg.generate_batch(cfg, nprob)
t = time.time(); xh, Sh, sth = pkg.phase1_batch(prob); th = time.time() - t
P, N, J = nprob, cfg.N, cfg.J
db = pkg.DeviceBatch(prob, np.zeros((P, N + J), dtype=np.int32), np.zeros((P, N)))
st = db.phase1(); torch.cuda.synchronize()
t = time.time()
for _ in range(3):
    st = db.phase1()
torch.cuda.synchronize(); tg = (time.time() - t) / 3
xg, Sg, stg = db.x0.cpu().numpy(), db.S0.cpu().numpy(), st.cpu().numpy()
print(name, nprob, "status eq", np.array_equal(stg, sth), "S eq", np.array_equal(Sg, Sh), "x0 eq", np.array_equal(xg, xh),
      "host %.1f ms (%.0f QPs/s)  gpu %.2f ms (%.0f QPs/s)" % (th * 1e3, nprob / th, tg * 1e3, nprob / tg), flush=True)
if not np.array_equal(Sg, Sh):
    bad = np.flatnonzero((Sg != Sh).any(axis=1))
    print("  bad problems", bad[:10], "first diffs", [(int(p), np.flatnonzero(Sg[p] != Sh[p])[:6].tolist()) for p in bad[:3]])
if not np.array_equal(xg, xh):
    bad = np.flatnonzero((xg != xh).any(axis=1))
    print("  x0 differs in", len(bad), "problems; max abs diff", np.abs(xg - xh).max())
