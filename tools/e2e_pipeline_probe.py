"""GPU box, experiment: end-to-end solveQP(Q) over the launch lanes with Phase-1 of step s + 1 DECOUPLED from the loop of step
s on the same lane -- its own stream, its own context, alternating (x0, S0) buffers -- against bench.py's round scheme (a
round's Phase-1 launches together, then its loops).  usage: e2e_pipeline_probe.py [steps per lane]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
import torch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NL, P = 3, 1024
cfg = pkg.CONFIGS["cfg4"]
dev = torch.device("cuda", 0)
lib = pkg._capi.lib()


class Lane:
    pass


lanes = []
for i in range(NL):
    ln = Lane()
    ln.ctx = pkg.Context(0)
    ln.ctx.set_option("lazy_handover", 1)
    ln.ctx.set_option("wave_qp_per_cu", 8)
    ln.p1ctx = pkg.Context(0)
    ln.batch, ln.prob, ln.x0h, ln.S0h = pkg.DeviceBatch.generated(cfg, P, pkg.BASE_SEED + i * P, ctx=ln.ctx, device=0)
    ln.batch.use_stats = False
    ln.stream, ln.p1stream = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    b = ln.batch
    ln.bufs = [(b.x0, b.S0), (torch.zeros_like(b.x0), torch.zeros_like(b.S0))]
    ln.p1status = torch.zeros(P, dtype=torch.int32, device=dev)
    lanes.append(ln)
torch.cuda.synchronize(dev)
cs = pkg.solver._csettings(None) if hasattr(pkg, "solver") else None
if cs is None:
    from importlib import import_module
    cs = import_module(pkg.__name__ + ".solver")._csettings(None)


def phase1_into(ln, which, stream):
    b = ln.batch
    x0, S0 = ln.bufs[which]
    rc = lib.ssqp_phase1_batch_dev_f64(ln.p1ctx.handle, b.P, b.N, b.M, b.J, *[b._ptr(b.t[k]) for k in "AGbgdu"], C.byref(cs),
                                       b._ptr(x0), b._ptr(S0), b._ptr(ln.p1status), C.c_void_p(stream.cuda_stream))
    pkg._capi.check(rc, ln.p1ctx.handle)


def reference_results():
    out = []
    for ln in lanes:
        with torch.cuda.stream(ln.stream):
            ln.batch.x0, ln.batch.S0 = ln.bufs[0]
            ln.batch.phase1()
            ln.batch.solve()
        ln.ctx.sync(ln.stream.cuda_stream)
    torch.cuda.synchronize(dev)
    for ln in lanes:
        r = ln.batch.results()
        out.append((r["S"].copy(), r["status"].copy()))
    return out


def rounds(n):
    """bench.py's scheme: a round's Phase-1 launches together, then its loops"""
    for _ in range(n):
        evs = []
        for ln in lanes:
            with torch.cuda.stream(ln.stream):
                ln.batch.x0, ln.batch.S0 = ln.bufs[0]
                ln.batch.phase1()
                e = torch.cuda.Event()
                e.record()
                evs.append(e)
        for ln in lanes:
            with torch.cuda.stream(ln.stream):
                for e in evs:
                    ln.stream.wait_event(e)
                ln.batch.solve()
    for ln in lanes:
        ln.ctx.sync(ln.stream.cuda_stream)
    torch.cuda.synchronize(dev)


def pipelined(n):
    """Phase-1 of step s + 1 on the lane's second stream while the loop of step s runs; (x0, S0) alternate"""
    pev = [[None] * (n + 1) for _ in lanes]
    dev_ev = [[None] * (n + 1) for _ in lanes]
    for li, ln in enumerate(lanes):
        phase1_into(ln, 0, ln.p1stream)
        pev[li][0] = torch.cuda.Event()
        pev[li][0].record(ln.p1stream)
    for s in range(n):
        for li, ln in enumerate(lanes):
            ln.stream.wait_event(pev[li][s])
            with torch.cuda.stream(ln.stream):
                ln.batch.x0, ln.batch.S0 = ln.bufs[s % 2]
                ln.batch.solve()
                dev_ev[li][s] = torch.cuda.Event()
                dev_ev[li][s].record()
            if s + 1 < n:
                if s >= 1:
                    ln.p1stream.wait_event(dev_ev[li][s - 1])      # (the buffers of step s + 1 were read by the loop of step s - 1)
                phase1_into(ln, (s + 1) % 2, ln.p1stream)
                pev[li][s + 1] = torch.cuda.Event()
                pev[li][s + 1].record(ln.p1stream)
    for ln in lanes:
        ln.ctx.sync(ln.stream.cuda_stream)
    torch.cuda.synchronize(dev)


ref = reference_results()
for name, fn in (("rounds (bench.py)", rounds), ("pipelined", pipelined), ("rounds (bench.py)", rounds), ("pipelined", pipelined)):
    fn(2)
    t = time.perf_counter()
    fn(steps)
    dt = time.perf_counter() - t
    same = True
    for ln, (S, st) in zip(lanes, ref):
        r = ln.batch.results()
        same = same and bool(np.array_equal(r["S"], S) and np.array_equal(r["status"], st))
    print("%-20s %d steps x %d lanes: %.3f ms per step -> %.0f QPs/s end to end; same S and status as one lane alone: %s" % (
        name, steps, NL, 1e3 * dt / (steps * NL), steps * NL * P / dt, same), flush=True)
