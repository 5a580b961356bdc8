"""Diagnostic (GPU box): distribution of K (free variables), W (kept constraint rows) and the event kinds over
the loop passes of a configuration, from the per-pass trace the kernel can write."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
nprob = int(sys.argv[2]) if len(sys.argv) > 2 else 128
db = pkg.DeviceBatch.generated(pkg.CONFIGS[name], nprob, ntrace=600)[0]
db.solve()
r = db.results()
it = r["status"]
tr = r["trace"]
K, W, kind = [], [], []
for p in range(nprob):
    n = min(int(it[p]), 600)
    K.append(tr[p, :n, 0]); W.append(tr[p, :n, 1]); kind.append(tr[p, :n, 2])
K = np.concatenate(K); W = np.concatenate(W); kind = np.concatenate(kind)
print("passes", len(K), "mean iters", it.mean())
print("K percentiles 10/50/90/99/max", np.percentile(K, [10, 50, 90, 99]), K.max())
print("W histogram", np.bincount(W))
print("kind histogram (0 freeK, 1 blocked, 2 release, 3 optimal)", np.bincount(kind))
print("max_k per problem percentiles", np.percentile(r["stats"]["max_k"], [50, 90, 99, 100]))
print("path bits", np.bincount(r["stats"]["path"]))
