"""Diagnostic (GPU box): where one loop pass spends its cycles.  Uses the -DSSQP_PHASE_PROFILE build
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
lib = pkg._capi.lib()
lib.ssqp_debug_phases.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 32)()
db.solve(); db.results()
lib.ssqp_debug_phases(out, 1)
db.solve(); res = db.results()
lib.ssqp_debug_phases(out, 1)
names = ["compaction+lists", "E-row sweep + Y copy", "rank filter (+barriers)", "hB pass V[:,nzB] + c   [old: pass1]",
         "forward border          [old: LDL]", "Schur + lambda", "v + back-substitution", "alpha/p/norm", "aStep",
         "gamma pass V[:,nz]", "KKTchk", "load/polish/store", "freeK", "factor sync (del/app)",
         "(sub) rank filter body / old LDL update", "(sub) old LDL panel"]
names += ["(sub) delete: rank-1 update", "(sub) delete: compaction", "(sub) append", "(sub) lambda solve", "(sub) aStep G rows",
          "(sub) sync: delete scan + deletes", "(sub) sync: append scan + appends", "", "#deletes", "#appends", "#aStep", "#gamma passes", "(wave 1) rank filter", "(wave 2) hB partial", "", ""]
tot = sum(out[:14])
iters = int(res["status"].sum())
print("config", name, "nprob", nprob, "total iterations", iters, "kernel ms", db.ctx.last_kernel_ms())
for i, n in enumerate(names):
    if 24 <= i < 28:
        if n: print("%-22s %10d  (%.3f per iteration)" % (n, out[i], out[i] / iters))
        continue
    if not n: continue
    print("%-22s %6.2f %%   %8.0f ticks/iter" % (n, 100.0 * out[i] / tot, out[i] / iters))
