import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
pkg = ge.load_package()
cfg = pkg.CONFIGS["cfg4"]
prob = pkg.generate_batch(cfg, 1024, nthreads=16)
x0, S0, st = pkg.phase1_batch(prob, nthreads=16)
pkg.solveQP_batch(prob, S0, x0)  # warm
ts = []
for _ in range(3):
    t = time.perf_counter(); z, S, status, detail = pkg.solveQP_batch(prob, S0, x0); ts.append(time.perf_counter() - t)
print("host-buffer path (PCIe copy of 2 GiB V included): %.1f ms per 1024-QP batch -> %.0f QPs/s" % (1e3 * min(ts), 1024 / min(ts)))
