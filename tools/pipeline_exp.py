"""Experiment (GPU box): K steps issued round-robin on two HIP streams (two contexts, separate outputs): the drain
of one batch (workgroups finishing at different times) overlaps the ramp-up of the next."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
cfg = pkg.CONFIGS["cfg4"]
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
lanes = []
for i in range(nl):
    ctx = pkg.Context(0)
    b = pkg.DeviceBatch.generated(cfg, 1024, pkg.BASE_SEED, ctx=ctx, device=0)[0]
    lanes.append((b, torch.cuda.Stream(dev)))
torch.cuda.synchronize(dev)
def run(k):
    for s in range(k):
        b, st = lanes[s % nl]
        with torch.cuda.stream(st):
            b.solve()
    torch.cuda.synchronize(dev)
run(nl)
t0 = time.perf_counter(); run(steps); el = time.perf_counter() - t0
print("lanes", nl, "steps", steps, "ms/step", 1e3 * el / steps, "QPs/s", 1024 * steps / el)
ref = lanes[0][0].results()
for b, _ in lanes[1:]:
    r = b.results()
    assert np.array_equal(r["S"], ref["S"]) and np.array_equal(r["status"], ref["status"])
print("outputs identical across lanes; all converged", bool((ref["status"] > 0).all()))
