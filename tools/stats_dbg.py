import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
cfg = pkg.CONFIGS["cfg4"]
prob = pkg.generate_batch(cfg, 4)
x0, S0, st = pkg.phase1_batch(prob)
db = pkg.DeviceBatch(prob, S0, x0, ntrace=512)
for mode in ("0", "1"):
    os.environ["SSQP_DENSE_GAMMA"] = mode
    db.solve(); r = db.results()
    print(mode, r["stats"], db.ctx.last_kernel_ms())
    tr = r["trace"][0][: r["status"][0]]
    K = tr[:, 0].astype(np.int64); kind = tr[:, 2]
    N = 512
    dense = (8 * (K * K + (N - K) * K) + 8 * 11 * N + 48 * N + 4 * (N + 10)).sum() + (8 * (N - K) ** 2)[kind >= 2].sum()
    print("expected dense-formula bytes problem 0:", dense)
