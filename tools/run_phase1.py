"""GPU box: run the Phase-1 kernel alone a few times (for rocprofv3 passes).  usage: run_phase1.py <cfg> <nprob> [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
nprob = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
db, prob, x0, S0 = pkg.DeviceBatch.generated(pkg.CONFIGS[name], nprob)
db.phase1(); db.torch.cuda.synchronize()
t = time.time()
for _ in range(reps):
    db.phase1()
db.torch.cuda.synchronize()
print("phase-1 %s x %d: %.3f ms per launch" % (name, nprob, (time.time() - t) / reps * 1e3))
