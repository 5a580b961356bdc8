"""Diagnostic (GPU box): cycles per phase of the wavefront-per-QP kernel.  Uses the -DSSQP_PHASE_PROFILE build
(make -C statusswitchingqp.jl_amd/csrc prof); shares only, never quote its run time."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SSQP_HIP_LIB"] = os.environ.get("SSQP_PROF_LIB") or os.path.join(ROOT, "statusswitchingqp.jl_amd", "libssqp_hip_prof.so")
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
nprob = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = pkg.CONFIGS[name]
prob = pkg.generate_batch(cfg, nprob)
x0, S0, st = pkg.phase1_batch(prob)
db = pkg.DeviceBatch(prob, S0, x0)
# optional context options as name=value (e.g. wave_kernel=2: start in the big-factor build, so every stamp is its own)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    db.ctx.set_option(k, int(v))
lib = pkg._capi.lib()
lib.ssqp_debug_wave_phases.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 64)()
db.solve(); db.results()
lib.ssqp_debug_wave_phases(out, 1)
db.solve(); res = db.results()
lib.ssqp_debug_wave_phases(out, 1)
names = ["rank filter", "Schur gather + lambda", "v + back substitution", "p, norm, accounting", "aStep ratios + min",
         "blocked: switches + bound shifts", "full step bookkeeping", "gamma pass", "KKT scan", "release + bound shift",
         "deletes (update, downdate, compaction, shifts)", "border sweep + H col after deletes", "c refresh before append",
         "one append (row, border row, H)", "  of which: gather V[F,j] + forward sweep + row stores",
         "fp32 screening of the bound columns (big-factor build; 'gamma pass' is then the rows + the exact candidates)"]
iters = int(res["status"].sum())
ms = db.ctx.last_kernel_ms()
tot = sum(out[:14]) + out[15]  # (slot 14 lies inside slot 13)
print("cycles per pass, whole QP lifetime / passes: %.0f" % (out[31] / iters))
print("config", name, "nprob", nprob, "total passes", iters, "kernel ms (diagnostic build)", ms)
print("stamped cycles per pass: %.0f" % (tot / iters))
for i, n in enumerate(names):
    cnt = out[32 + i]
    print("%-48s %6.2f %%  %8.0f cyc/pass  %8.0f cyc/occurrence  (%.3f per pass)" % (
        n, 100.0 * out[i] / tot, out[i] / iters, out[i] / max(cnt, 1), cnt / iters))
