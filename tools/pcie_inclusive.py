"""GPU box: the host-buffer entry points (what a Julia ccall with host arrays uses), PCIe copy of V included, and the
device-resident problem handle.  Never bench.py's `value` (that is the hot path from HBM-resident inputs)."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
from oracle import oracle as orc
cfg = pkg.CONFIGS["cfg4"]
prob = pkg.generate_batch(cfg, 1024, nthreads=16)
x0, S0, st = pkg.phase1_batch(prob, nthreads=16)
z0, Sr, st0, _ = pkg.solveQP_batch(prob, S0, x0)  # warm
ts = []
for _ in range(3):
    t = time.perf_counter(); z, S, status, detail = pkg.solveQP_batch(prob, S0, x0); ts.append(time.perf_counter() - t)
vb = prob["V"].nbytes
print("ssqp_solve_batch_f64 (host buffers, %.2f GiB of V over PCIe, chunked + 4 launch lanes): %.1f ms per 1024-QP batch -> %.0f QPs/s; "
      "PCIe bound at 63 GB/s = %.1f ms" % (vb / 2**30, 1e3 * min(ts), 1024 / min(ts), vb / 63e9 * 1e3))
assert np.array_equal(S, Sr) and np.array_equal(status, st0) and np.array_equal(z, z0)
ctx = pkg.default_context()
ctx.set_option("pin_host_buffers", 1)
t = time.perf_counter(); pkg._capi  # first call registers V
V = np.ascontiguousarray(prob["V"])
prob["V"] = V
t = time.perf_counter(); z, S, status, detail = pkg.solveQP_batch(prob, S0, x0); t_first = time.perf_counter() - t
ts = []
for _ in range(3):
    t = time.perf_counter(); z, S, status, detail = pkg.solveQP_batch(prob, S0, x0); ts.append(time.perf_counter() - t)
print("  with pin_host_buffers=1: first call (registers V) %.1f ms, then %.1f ms per batch -> %.0f QPs/s = %.2f x the PCIe bound"
      % (1e3 * t_first, 1e3 * min(ts), 1024 / min(ts), min(ts) / (vb / 63e9)))
assert np.array_equal(S, Sr) and np.array_equal(status, st0) and np.array_equal(z, z0)
ctx.set_option("pin_host_buffers", 0)
rb = pkg.ResidentBatch(prob)
rb.solve(S0, x0)
ts = []
for _ in range(5):
    t = time.perf_counter(); z2, S2, status2, _ = rb.solve(S0, x0); ts.append(time.perf_counter() - t)
print("ssqp_problem_solve (V resident, only S/x0 in and z/S/status out): %.2f ms per batch -> %.0f QPs/s" % (1e3 * min(ts), 1024 / min(ts)))
assert np.array_equal(S2, Sr) and np.array_equal(status2, st0) and np.array_equal(z2, z0)
zo, So, sto, _, _ = orc.solveQP_warm_batch(*[prob[k][:64] for k in "VAGqbgdu"], S0[:64], x0[:64])
assert np.array_equal(S[:64], So) and np.array_equal(status[:64], sto)
print("results identical across the three entry points; first 64 match the oracle")
