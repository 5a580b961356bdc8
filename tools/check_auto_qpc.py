"""GPU box: which wavefront build the by-batch-size rule ends up with (hand-over counts of consecutive batches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
nprob = int(sys.argv[2]) if len(sys.argv) > 2 else 1100
ctx = pkg.Context(0)
b, prob, x0, S0 = pkg.DeviceBatch.generated(pkg.CONFIGS[name], nprob, ctx=ctx)
for rep in range(4):
    b.solve()
    r = b.results()
    st = r["stats"]
    print("batch %d: handed over %d of %d, kernel ms %.3f, converged %d" % (
        rep, int(((st["path"] & 32) != 0).sum()), nprob, ctx.last_kernel_ms(), int((r["status"] > 0).sum())))
