#!/bin/bash
# GPU box: four vs eight QPs per CU for the wavefront kernel (parity, serial and pipelined throughput)
for q in 4 8; do
  python tools/debug_parity.py cfg4 1024 0 wave_qp_per_cu=$q 2>&1 | grep -v amdgpu | cut -c1-12,60-230
  python tools/debug_parity.py cfg4 4096 0 wave_qp_per_cu=$q 2>&1 | grep -v amdgpu | cut -c1-12,60-230
done
python - <<'PY'
import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
pkg = ge.load_package()
import torch
cfg = pkg.CONFIGS["cfg4"]
for q in (4, 8):
    ctxs = [pkg.Context(0) for _ in range(3)]
    for c in ctxs:
        c.set_option("wave_qp_per_cu", q); c.set_option("lazy_handover", 1)
    b0, prob, x0, S0 = pkg.DeviceBatch.generated(cfg, 1024, ctx=ctxs[0])
    lanes = [(b0, torch.cuda.current_stream())] + [(b0.twin(c), torch.cuda.Stream()) for c in ctxs[1:]]
    def run(n):
        for i in range(n):
            b, st = lanes[i % 3]
            with torch.cuda.stream(st):
                b.solve()
        for b, st in lanes:
            b.ctx.sync(st.cuda_stream)
        torch.cuda.synchronize()
    run(6)
    t = time.perf_counter(); run(18); dt = (time.perf_counter() - t) / 18
    r = [b.results() for b, _ in lanes]
    same = all(np.array_equal(x["S"], r[0]["S"]) and np.array_equal(x["status"], r[0]["status"]) for x in r)
    print("qpc", q, "3 lanes: %.3f ms/step -> %.0f QPs/s  lanes agree %s  converged %s" % (dt * 1e3, 1024 / dt, same, bool((r[0]["status"] > 0).all())), flush=True)
PY
