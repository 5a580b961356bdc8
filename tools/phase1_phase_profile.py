"""Diagnostic (GPU box): cycles per phase of the Phase-1 kernel (thread 0 of every workgroup).  Uses the
-DSSQP_PHASE_PROFILE build (make -C statusswitchingqp.jl_amd/csrc prof); shares only, never quote its run time."""
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
db, prob, x0, S0 = pkg.DeviceBatch.generated(cfg, nprob)
lib = pkg._capi.lib()
lib.ssqp_debug_phase1_phases.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 24)()
db.ctx.set_option("phase1_wave", 0)
db.phase1(); db.torch.cuda.synchronize()
lib.ssqp_debug_phase1_phases(out, 1)
db.phase1(); db.torch.cuda.synchronize()
lib.ssqp_debug_phase1_phases(out, 1)
names = ["set-up (LP assembly, first Y)", "pricing", "block maximum", "entering column invB a_k", "ratio test (thread 0)",
         "basis sort + gather", "inv(lu(B))", "Y = invB A1", "xb = invB b - Y x"]
tot = sum(out[:9])
it, bc = out[14], out[15]
print("config", name, "nprob", nprob, "simplex passes per QP %.1f, basis changes per QP %.1f, cycles per QP %.0f" % (
    it / nprob, bc / nprob, tot / nprob))
for i, n in enumerate(names):
    print("%-34s %6.2f %%  %9.0f cycles per QP" % (n, 100.0 * out[i] / tot, out[i] / nprob))
if out[10] or out[11]:
    print("  inv(lu(B)) of the many-rows build: elimination %.0f, columns of the inverse %.0f cycles per basis change" % (
        out[10] / max(bc, 1), out[11] / max(bc, 1)))
    print("  elimination per basis change: %.1f steps with a nonzero L column, %.1f runs with moved rows; wavefront 0 alone %.0f cycles, workgroup phases %.0f" % (
        out[16] / max(bc, 1), out[19] / max(bc, 1), out[17] / max(bc, 1), out[18] / max(bc, 1)))
    print("  Y.c refresh of the many-rows build, column loop per refresh: group set-up + first loads %.0f, step blocks %.0f, row sums %.0f cycles" % (
        out[9] / max(bc + 1, 1), out[12] / max(bc + 1, 1), out[13] / max(bc + 1, 1)))

# ---- the one-wavefront-per-QP kernel (the default where it applies): its own stamps
lib.ssqp_debug_phase1_wave_phases.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
db.ctx.set_option("phase1_wave", 1)
db.phase1(); db.torch.cuda.synchronize()
lib.ssqp_debug_phase1_wave_phases(out, 1)
db.phase1(); db.torch.cuda.synchronize()
lib.ssqp_debug_phase1_wave_phases(out, 1)
wn = ["set-up (LP columns into registers, first Y.c)", "pricing + first maximum", "entering column + ratio test",
      "basis sort + columns", "inv(lu(B)) in registers", "rows of the inverse, statuses", "Y.c refresh", "xb: cached terms re-added, invB b", "bound flip + its cached term", "xb terms anew after a pivot"]
tot = sum(out[:10])
if tot:
    print("wavefront kernel: simplex passes per QP %.1f, basis changes per QP %.1f, cycles per QP %.0f" % (
        out[14] / nprob, out[15] / nprob, tot / nprob))
    print("xb rebuilds per QP %.1f, listed columns per rebuild %.1f" % (out[13] / nprob, out[12] / max(out[13], 1)))
    for i, n in enumerate(wn):
        print("%-46s %6.2f %%  %9.0f cycles per QP" % (n, 100.0 * out[i] / tot, out[i] / nprob))
