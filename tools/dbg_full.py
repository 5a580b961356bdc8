"""GPU box: the single-launch solveQP(Q) against the two-launch path (same bits?) and their times.  usage: dbg_full.py cfg nprob"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
import torch
name, nprob = sys.argv[1], int(sys.argv[2])
cfg = pkg.CONFIGS[name]
prob = pkg.generate_batch(cfg, nprob, with_V=False)
P, N, J = nprob, cfg.N, cfg.J
db = pkg.DeviceBatch(prob, np.zeros((P, N + J), dtype=np.int32), np.zeros((P, N)), _gen=(cfg, pkg.BASE_SEED))
def timed(fn, n=5):
    fn(); db.ctx.sync(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); db.ctx.sync(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def two():
    db.phase1(); db.solve()
t2 = timed(two); r2 = db.results()
t1 = timed(db.solve_full); r1 = db.results()
print(name, nprob, "two launches %.3f ms (%.0f QPs/s)   one launch %.3f ms (%.0f QPs/s)   same S %s status %s z %s  conv %s" % (
    t2, P / t2 * 1e3, t1, P / t1 * 1e3, np.array_equal(r1["S"], r2["S"]), np.array_equal(r1["status"], r2["status"]),
    np.array_equal(r1["z"], r2["z"]), bool((r1["status"] > 0).all())), flush=True)
