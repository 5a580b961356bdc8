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
for wave in ([1, 0] if len(sys.argv) < 4 else [int(sys.argv[3])]):
  db.ctx.set_option("phase1_wave", wave)
  db.x0.zero_(); db.S0.zero_()
  st = db.phase1(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(5):
    st = db.phase1()
  e1.record()
  torch.cuda.synchronize(); tg = e0.elapsed_time(e1) * 1e-3 / 5
  xg, Sg, stg = db.x0.cpu().numpy(), db.S0.cpu().numpy(), st.cpu().numpy()
  print("phase1_wave =", wave, name, nprob, "status eq", np.array_equal(stg, sth), "S eq", np.array_equal(Sg, Sh), "x0 eq", np.array_equal(xg, xh),
      "host %.1f ms (%.0f QPs/s)  gpu %.2f ms (%.0f QPs/s)" % (th * 1e3, nprob / th, tg * 1e3, nprob / tg), flush=True)
  if not np.array_equal(stg, sth):
    print("  status gpu", stg[:16], "host", sth[:16])
  if not np.array_equal(Sg, Sh):
    bad = np.flatnonzero((Sg != Sh).any(axis=1))
    print("  bad problems", len(bad), bad[:10], "first diffs", [(int(p), np.flatnonzero(Sg[p] != Sh[p])[:6].tolist()) for p in bad[:3]])
  if not np.array_equal(xg, xh):
    bad = np.flatnonzero((xg != xh).any(axis=1))
    print("  x0 differs in", len(bad), "problems; max abs diff", np.abs(xg - xh).max())
