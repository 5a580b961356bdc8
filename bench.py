#!/usr/bin/env python3
"""bench.py -- QPs/sec of the in-kernel active-set loop on batched N=512 dense portfolio QPs.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (solveQP(Q,S,x0), SSQP.jl:237-377) over one batch of
synthetic QPs already resident in HBM: BASELINE.json configs[3] ("8192 independent N=512 QPs
sharded across 8 GPUs") = 1024 QPs per GPU, cfg2-style problems (M=1, J=10, box bounds).
Weak scaling: every rank owns 1024 problems; the only collective is the final RCCL all-gather of
(z, S, status), inside the timed region, issued from ONE communication stream.  Rank 0 prints ONE JSON line.

Consecutive steps are independent batches: every launch lane owns a DISTINCT batch (own seeds, own V in HBM).  In
mode "lanes" the steps are issued round-robin on `--streams` HIP streams (default 3, one library context per lane), so
the drain of one launch -- QPs need 143..262 passes, the launch ends with the slowest -- overlaps the start of the
next; in mode "serial" the same batches run one after the other on one stream.  Both are probed after warm-up and the
faster one is the timed configuration (`--mode` forces one).  The top-level `roofline` describes the kernel build and
launch mode `value` was timed on, with launch durations from HIP events over the timed region; `roofline_serial` keeps
the single-launch figures.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# vector-instruction issue: 256 CUs x 4 SIMDs at 2.4 GHz, one wave64 f64 VALU instruction per 4 cycles per SIMD (16 f64
# lanes per cycle = the 78.6 TFLOP/s vector-f64 figure; the SQ counters agree: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.05
# quad-cycles per instruction for this kernel)
VALU_ISSUE_PER_CU = 4 * 2.4e9 / 4.0   # x the device's CU count (256 on MI355X), read from the device at run time
KERNEL_SOURCES = ["ssqp_wave.hip", "ssqp_kernels.hip", "ssqp_device.h", "ssqp_internal.h", "ssqp_api.hip",
                  "ssqp_phase1.hip", "ssqp_host.cpp", "../../include/ssqp_hip.h", "ssqp_phase1_wave.h", "ssqp_phase1_wave.hip"]


def kernel_source_hash():
    """sha256 over the library's sources (csrc/Makefile: HASHED, same order): ssqp_version() of a binary built from
    them ends in this hash, and PMC figures under profiles/ are only quoted for the code they were taken on"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "statusswitchingqp.jl_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nprob", type=int, default=1024, help="QPs per GPU")
    ap.add_argument("--config", default="cfg4", help="problem family (statusswitchingqp.jl_amd CONFIGS)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target CPU time of each cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--streams", type=int, default=3,
                    help="launch lanes: consecutive steps go round-robin to this many HIP streams (1: serial launches)")
    ap.add_argument("--mode", default="auto", choices=["auto", "lanes", "serial"],
                    help="auto: probe both launch modes after warm-up and time the faster one")
    ap.add_argument("--skip-dense", action="store_true",
                    help="do not time the dense-formulation launches (profiling runs want one kernel variant)")
    ap.add_argument("--dense", action="store_true", help="time the dense (reference-shaped) formulation as the main run")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps steps: the first is the contractual one (`value`), the others give "
                         "the median (SURVEY.md 8d: median of >= 5 repetitions)")
    ap.add_argument("--keep-stats", action="store_true",
                    help="time the accounting builds (statistics pointer passed) instead of the lean ones: A/B runs")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="a context option of the library (include/ssqp_hip.h: wave_kernel, wave_qp_per_cu, ...) set on every "
                         "lane on top of the mode's own; recorded in config.options, and the PMC file (collected with the "
                         "defaults) is then not quoted")
    ap.add_argument("--pmc-json", default=os.path.join(ROOT, "profiles", "pmc_counters.json"),
                    help="per-launch PMC figures (HBM bytes, SQ counters) from separate rocprofv3 --pmc passes")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as FRESH child processes (this process
    has not touched the GPU, and never replaces itself) and relay rank 0's JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    if world_env is not None and int(world_env) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s\n" % (args.gpus, world_env))
        raise SystemExit(2)
    run(args)


def run(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the multi-rank path on a one-GPU box: every rank on device 0, gloo instead of RCCL
    rehearsal = os.environ.get("SSQP_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    num_cu = int(torch.cuda.get_device_properties(dev).multi_processor_count)
    valu_issue_peak = num_cu * VALU_ISSUE_PER_CU
    cfg = pkg.CONFIGS[args.config]
    ncpu = usable_cores()
    gen_threads = max(1, ncpu // max(1, min(world, 8)))
    t0 = time.time()
    # ---- launch lanes: ONE DISTINCT BATCH PER LANE (own seeds, own V in HBM, own context = workspace + work counters,
    # own outputs).  Consecutive steps are independent batches; in "lanes" mode they are issued round-robin on the
    # lanes' HIP streams so that the drain of one launch (QPs need 143..262 passes, a launch ends with its slowest)
    # overlaps the start of the next; in "serial" mode the same batches are solved one after the other on one stream.
    nlanes = max(1, args.streams)
    main_stream = torch.cuda.current_stream(dev)

    class Lane:
        pass
    lanes = []
    for i in range(nlanes):
        ln = Lane()
        ln.ctx = pkg.Context(local)
        if args.dense:
            ln.ctx.set_option("dense_gamma", 1)
        # V (the N*N*T part) is generated on the GPU, bit-identical to the host generator; the small arrays and the
        # Phase-1 vertex (x0, S0) come from the host C++ (not timed: the metric is the hot path solveQP(Q,S,x0))
        seed0 = pkg.BASE_SEED + (rank * nlanes + i) * args.nprob
        ln.batch, ln.prob, ln.x0, ln.S0 = pkg.DeviceBatch.generated(cfg, args.nprob, seed0, ctx=ln.ctx, device=local,
                                                                   nthreads=gen_threads)
        ln.stream = main_stream if i == 0 else torch.cuda.Stream(dev)
        ln.seed0 = seed0
        lanes.append(ln)
    torch.cuda.synchronize(dev)
    t_setup = time.time() - t0
    batch = lanes[0].batch
    P, N, J = batch.P, batch.N, batch.J
    comm = torch.cuda.Stream(dev) if world > 1 else None   # the ONE stream every gather is issued from
    # the final gather: one pre-allocated receive buffer per lane, ONE collective per step (dist.PackedGather; the
    # send side is the batch's packed output buffer itself)
    for ln in lanes:
        ln.pg = pkg.dist.PackedGather(P, N, J, dev, world) if world > 1 else None

    # wavefronts per CU of the wavefront kernel: with several batches in flight two per SIMD (8 per CU) give more QPs/s;
    # a single 1024-QP launch fills 4 per CU exactly and runs fastest with one per SIMD (include/ssqp_hip.h)
    MODES = {"lanes": dict(lazy_handover=1, wave_qp_per_cu=8), "serial": dict(lazy_handover=0, wave_qp_per_cu=0)}
    state = dict(mode="serial", step=0, gathers=0, coll_allocs=0)

    def n_allocs():
        return torch.cuda.memory_stats(dev).get("allocation.all.allocated", 0)
    owed = [False] * nlanes                 # lanes whose last solve has not been gathered yet

    def set_mode(mode):
        state["mode"] = mode
        for ln in lanes:
            ln.ctx.sync(ln.stream.cuda_stream)
            for k, v in MODES[mode].items():
                ln.ctx.set_option(k, v)
            for kv in args.option:                      # (the caller's own choices win over the mode's)
                k, v = kv.split("=", 1)
                ln.ctx.set_option(k, int(v))
        torch.cuda.synchronize(dev)

    def lane_stream(i):
        return lanes[i].stream if state["mode"] == "lanes" else main_stream

    def gather(i):
        """final gather of lane i's step (RCCL over xGMI), issued from the one communication stream in launch order;
        the lane's next solve waits for it (it overwrites the buffer the gather sends)"""
        if not owed[i]:
            return
        ln = lanes[i]
        ln.ctx.flush()                      # (lazy hand-over: an owed workgroup-kernel launch goes out first)
        st = lane_stream(i)
        comm.wait_stream(st)
        a0 = n_allocs()
        with torch.cuda.stream(comm):
            ln.pg.gather(ln.batch.out)
        state["coll_allocs"] += n_allocs() - a0     # (what the backend allocates inside the collective call itself)
        st.wait_stream(comm)
        owed[i] = False
        state["gathers"] += 1

    def step(events=None):
        i = state["step"] % nlanes
        ln = lanes[i]
        state["step"] += 1
        if world > 1:
            gather(i)                       # the step this lane ran last time round
        with torch.cuda.stream(lane_stream(i)):
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            ln.batch.solve()                # in-kernel active-set loop, asynchronous on the lane's stream
            if events is not None:
                e1.record()
                events.append((e0, e1))
        owed[i] = world > 1

    def fence():
        if world > 1:
            first = state["step"] % nlanes  # oldest first: launch order
            for k in range(nlanes):
                gather((first + k) % nlanes)
        for i, ln in enumerate(lanes):      # (lazy hand-over: every lane's last launch is settled)
            ln.ctx.sync(lane_stream(i).cuda_stream)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(nsteps, events=None):
        fence()
        state["step"] = 0
        state["gathers"] = 0
        state["coll_allocs"] = 0
        t = time.perf_counter()
        for _ in range(nsteps):
            step(events)
        fence()
        return time.perf_counter() - t

    # ---- which mode?  Both are warmed and probed on a few steps; the faster one is the timed configuration (lanes do not
    # help a workload whose QPs are all handed to a kernel that already fills the chip).  --mode forces one.
    probe = {}
    modes = ["serial"] if nlanes == 1 else ["lanes", "serial"]
    if args.mode != "auto":
        modes = [args.mode if nlanes > 1 else "serial"]
    for m in modes:
        set_mode(m)
        timed(max(args.warmup, nlanes))                      # (every lane is warmed once in this mode)
        if len(modes) > 1:
            n = max(2, min(args.steps, 2 * nlanes))
            probe[m] = timed(n) / n
    mode = modes[0] if len(modes) == 1 else min(probe, key=probe.get)
    if world > 1 and len(modes) > 1:                         # (one choice for the whole job)
        flag = torch.tensor([1 if mode == "lanes" else 0], dtype=torch.int32, device=dev if not rehearsal else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        mode = "lanes" if int(flag.item()) == 1 else "serial"
    set_mode(mode)
    qpc_timed = 8 if mode == "lanes" else (8 if P > 4 * num_cu else 4)

    # ---- one ACCOUNTING launch per lane in the timed configuration (statistics on: the bytes the formulation reads, passes,
    # which kernels ran), then the timed region with the statistics pointer NULL -- what a production caller passes: the
    # library then runs the kernel builds that do not carry the accounting (same decisions; S / status compared below)
    timed(nlanes)
    acct = [ln.batch.results() for ln in lanes]
    for ln in lanes:
        ln.batch.use_stats = bool(args.keep_stats)
    timed(nlanes)                                            # (the lean builds' first launch: module load, outside the clock)
    # ---- the timed region
    timed(0)                                                 # (settle: the counters below cover the timed steps only)
    alloc0 = n_allocs()
    events = []
    elapsed = timed(args.steps, events)
    alloc1 = n_allocs()
    gathers_timed = state["gathers"]
    assert world == 1 or state["gathers"] == args.steps, (state["gathers"], args.steps)   # one gather per timed step, all inside
    launch_ms = [e0.elapsed_time(e1) for e0, e1 in events]   # whole solve() calls: S reset, counters, prep kernel, solve kernels
    # the solve kernels alone: HIP events the library records around them on the stream of each launch (it keeps the
    # pairs of the last 16 launches per context) -- every timed launch, or the most recent 16 per lane
    kern_ms = []
    for i, ln in enumerate(lanes):
        n_i = len(range(i, args.steps, nlanes))
        if n_i:
            kern_ms += ln.ctx.recent_kernel_ms(min(n_i, 16))
    k_ms = float(np.mean(kern_ms))
    in_flight = k_ms * args.steps * 1e-3 / elapsed           # launches in flight on average (lanes overlap)
    last_kernel_ms = lanes[(args.steps - 1) % nlanes].ctx.last_kernel_ms()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if not rehearsal else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # further repetitions of the same region (same steps, same fences): `value` stays the first, contractual region;
    # the median over all of them is reported beside it
    rep_s = [elapsed]
    for _ in range(max(0, args.repeats - 1)):
        e = timed(args.steps)
        if world > 1:
            tt = torch.tensor([e], dtype=torch.float64, device=dev if not rehearsal else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e = float(tt.item())
        rep_s.append(e)

    used = lanes[:min(nlanes, args.steps)]
    timed_res = [ln.batch.results() for ln in lanes]
    lean_same = all(bool(np.array_equal(t["S"], a["S"]) and np.array_equal(t["status"], a["status"]) and np.array_equal(t["z"], a["z"]))
                    for t, a in zip(timed_res[:len(used)], acct[:len(used)]))
    results = acct                                           # (statistics: from the accounting launches)
    res = results[0]
    ok = all(bool((r["status"] > 0).all()) for r in results[:len(used)])
    distinct = nlanes == 1 or not np.array_equal(results[0]["S"], results[1]["S"])
    stats = res["stats"]
    # bytes this kernel's formulation has to read, per launch: mean over the lanes' batches
    read_bytes = float(np.mean([r["stats"]["read_bytes"].sum() for r in results[:len(used)]]))
    dense_bytes = int(stats["alg_bytes"].sum())     # bytes of the reference's dense formulation
    iters = np.concatenate([r["status"].astype(np.int64) for r in results[:len(used)]])
    wave_share = float(((stats["path"] & 16) != 0).mean())
    handed_over = int(((stats["path"] & 32) != 0).sum())
    big_wave = int(((stats["path"] & 64) != 0).sum())
    mean_maxk = float(stats["max_k"].mean())

    # ---- serial single launches of lane 0's batch, each timed by the library's own HIP events (the kernel alone)
    ctx = lanes[0].ctx

    def timed_launches(n):
        ms = []
        for _ in range(n):
            batch.solve()
            ctx.sync(main_stream.cuda_stream)
            torch.cuda.synchronize(dev)
            ms.append(ctx.last_kernel_ms())
        return ms
    set_mode("serial")
    batch.use_stats = True
    timed_launches(1)
    res_serial = batch.results()                             # (accounting launch of the serial configuration)
    batch.use_stats = bool(args.keep_stats)
    iso = timed_launches(5)[1:]
    iso_ms = float(np.mean(iso))
    res_lean = batch.results()
    lean_same = lean_same and bool(np.array_equal(res_lean["S"], res_serial["S"]) and np.array_equal(res_lean["status"], res_serial["status"]))
    read_serial = int(res_serial["stats"]["read_bytes"].sum())
    same_serial = bool(np.array_equal(res_serial["S"], res["S"]) and np.array_equal(res_serial["status"], res["status"]))
    # end-to-end solveQP(Q) = initQP + loop (SSQP.jl:224-234) with BOTH stages on the GPU, serial launches: Phase-1
    # kernel (bit-identical to the host stage that produced the resident vertex), then the loop.  Labelled, never
    # the headline: the metric is the hot path from a resident vertex.
    e2e = None
    if not args.dense:
        try:
            batch.phase1()
            torch.cuda.synchronize(dev)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ts = time.perf_counter()
            for i in range(args.steps):
                if i == 0:
                    ev[0].record()
                batch.phase1()
                if i == 0:
                    ev[1].record()
                batch.solve()
                if i == 0:
                    ev[2].record()
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - ts) / args.steps
            r2 = batch.results()
            e2e = {"qps": P / dt, "ms_per_step": 1e3 * dt, "phase1_ms": ev[0].elapsed_time(ev[1]),
                   "loop_ms": ev[1].elapsed_time(ev[2]),
                   "same_S_and_iters": bool(np.array_equal(r2["S"], res["S"]) and np.array_equal(r2["status"], res["status"])),
                   "note": "ssqp_phase1_batch_dev_f64 (one wavefront per QP where that kernel applies) + the loop per step, "
                           "two launches on one stream"}
            try:   # the same in ONE launch per QP (ssqp_solve_full_batch_dev_f64), where that entry takes the shape
                batch.solve_full()
                ctx.sync(main_stream.cuda_stream)
                torch.cuda.synchronize(dev)
                ts = time.perf_counter()
                for i in range(args.steps):
                    batch.solve_full()
                ctx.sync(main_stream.cuda_stream)
                torch.cuda.synchronize(dev)
                dt1 = (time.perf_counter() - ts) / args.steps
                r4 = batch.results()
                e2e["single_launch"] = {"qps": P / dt1, "ms_per_step": 1e3 * dt1,
                                        "same_S_and_iters": bool(np.array_equal(r4["S"], res["S"]) and
                                                                 np.array_equal(r4["status"], res["status"]))}
            except Exception as exc:
                e2e["single_launch"] = {"error": str(exc)}
            if nlanes > 1 and world == 1 and mode == "lanes":
                # the same with the launch lanes: Phase-1 and loop of a step on its lane's stream, so one step's
                # Phase-1 runs beside another step's loop
                set_mode("lanes")

                def e2e_steps(n):
                    # A round = one step per lane.  The Phase-1 kernel keeps a QP's whole LP in 512 registers per lane: one
                    # wavefront per SIMD and nothing beside it, so a Phase-1 launch that trickles in between another lane's
                    # loop wavefronts (two per SIMD) stalls both.  The round's Phase-1 launches therefore go out together, on
                    # their lanes' streams, and the round's loops start once all of them are done (events): Phase-1 fills the
                    # SIMDs the previous round's loops drain, then the loops share the chip among themselves as in the timed
                    # region.
                    for r0 in range(0, n, nlanes):
                        rl = [lanes[i % nlanes] for i in range(r0, min(n, r0 + nlanes))]
                        evs = []
                        for ln in rl:
                            with torch.cuda.stream(ln.stream):
                                ln.batch.phase1()
                                e = torch.cuda.Event()
                                e.record()
                                evs.append(e)
                        for ln in rl:
                            with torch.cuda.stream(ln.stream):
                                for e in evs:
                                    ln.stream.wait_event(e)
                                ln.batch.solve()
                    for ln in lanes:
                        ln.ctx.sync(ln.stream.cuda_stream)
                    torch.cuda.synchronize(dev)
                e2e_steps(nlanes)
                ts = time.perf_counter()
                e2e_steps(args.steps)
                dtl = (time.perf_counter() - ts) / args.steps
                same = True
                for ln, r0 in zip(lanes, results):
                    r3 = ln.batch.results()
                    same = same and bool(np.array_equal(r3["S"], r0["S"]) and np.array_equal(r3["status"], r0["status"]))
                e2e["lanes"] = {"streams": nlanes, "qps": P / dtl, "ms_per_step": 1e3 * dtl, "same_S_and_iters": same}
                set_mode("serial")
        except Exception as exc:   # (e.g. M + J too large for the GPU Phase-1)
            e2e = {"error": str(exc)}
    # the same problem through the workgroup kernel with the gamma pass reading EVERY column of V, as the reference's
    # dense V[B,F]*alpha + V[B,B]*zB does (SSQP.jl:352): the HBM-bound formulation, timed beside the default one
    dense = None
    if not args.skip_dense and not args.dense:
        batch.use_stats = True                               # (the workgroup kernel: its byte count is the point of this leg)
        with ctx.options(dense_gamma=1):
            dense_ms = float(np.mean(timed_launches(2)))
            res_dense = batch.results()
        dense = {"ms": dense_ms, "read": int(res_dense["stats"]["read_bytes"].sum()),
                 "same": bool(np.array_equal(res_dense["S"], res["S"]) and
                              np.array_equal(res_dense["status"], res["status"]))}

    # PMC figures are quoted only when they were collected on exactly this kernel source and workload
    pmc = None
    if os.path.exists(args.pmc_json) and not args.option:
        try:
            with open(args.pmc_json) as f:
                pj = json.load(f)
            if (pj.get("config") == args.config and pj.get("nprob") == args.nprob and
                    pj.get("kernel_source_sha256") == kernel_source_hash()):
                pmc = pj
        except Exception:
            pmc = None

    def pmc_of(key):
        return pmc[key] if pmc and key in pmc else {}
    key_serial = "dense_formulation" if args.dense else "default_formulation"
    key_timed = "eight_per_cu_build" if (qpc_timed == 8 and not args.dense) else key_serial
    traffic = pmc_of(key_timed).get("hbm_bytes_per_launch")
    sq = pmc_of(key_timed).get("sq")
    traffic_serial = pmc_of(key_serial).get("hbm_bytes_per_launch")
    sq_serial = pmc_of(key_serial).get("sq")
    traffic_dense = pmc_of("dense_formulation").get("hbm_bytes_per_launch")

    def issue_block(sqc, ms, iters_sum, what):
        if not sqc or "SQ_INSTS_VALU" not in sqc:
            return None
        return {"bound": "valu-issue", "achieved": sqc["SQ_INSTS_VALU"] / (ms * 1e-3), "peak": valu_issue_peak,
                "unit": "wave-instructions/s", "frac": sqc["SQ_INSTS_VALU"] / (ms * 1e-3) / valu_issue_peak,
                "issue_active_frac": sqc["SQ_ACTIVE_INST_ANY"] / sqc["SQ_WAVE_CYCLES"],
                "wait_frac": sqc["SQ_WAIT_ANY"] / sqc["SQ_WAVE_CYCLES"],
                "valu_per_pass": sqc["SQ_INSTS_VALU"] / iters_sum, "salu_per_pass": sqc["SQ_INSTS_SALU"] / iters_sum,
                "note": "SQ counters per launch of " + what + " from profiles/ (same source hash, separate --pmc passes) / "
                        "the time one launch's worth of work takes here; peak = %d CUs x 4 SIMDs x 2.4 GHz / 4 cycles "
                        "per wave64 f64 VALU instruction" % num_cu}

    out = None
    if rank == 0:
        qps = world * P * args.steps / elapsed
        kname_wave = "ssqp_wave_kernel<%s, true>" % ("2, true, 2" if qpc_timed == 8 else "1, false, 2")
        if wave_share > 0 and handed_over > P // 2:
            kname = "ssqp_solve_kernel (QPs handed over by the wavefront kernel)"
        elif wave_share > 0 and big_wave > P // 2:
            kname = "ssqp_wave_kernel<1, false, 4, true> (big-factor build, after %s)" % kname_wave
        elif wave_share > 0:
            kname = kname_wave
        else:
            kname = "ssqp_solve_kernel"
        conc = max(1.0, in_flight)
        achieved = read_bytes * conc / (k_ms * 1e-3) / 1e9
        hot_set = qpc_timed * num_cu * mean_maxk * N * 8.0
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "achieved_is": "ALGORITHMIC bytes per second: what the formulation reads, counted in-kernel, L2 and "
                               "Infinity-Cache hits included -- an upper bound on the DRAM share, not a DRAM rate",
                "kernel": kname, "mode": mode, "wave_qp_per_cu": qpc_timed,
                "kernel_ms": k_ms, "launches_in_flight": in_flight, "alg_bytes_per_launch": read_bytes,
                "achieved_from_traffic": None if traffic is None else traffic * conc / (k_ms * 1e-3) / 1e9,
                "frac_from_traffic": None if traffic is None else traffic * conc / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "hot_set_in_flight_bytes": hot_set,
                "note": "the kernel and launch mode `value` was timed on.  kernel_ms = average duration of the solve "
                        "kernel over the launches of the timed region (HIP events the library records around it on the "
                        "stream each launch went to); achieved = alg_bytes_per_launch x "
                        "max(1, launches_in_flight) / kernel_ms, launches_in_flight = sum of launch durations / wall time "
                        "(launch lanes overlap: a launch shares the chip with its neighbours, so bytes / its own duration "
                        "would understate the rate the chip moves bytes at; with serial launches the factor is 1).  "
                        "alg_bytes = bytes this formulation reads, counted in-kernel (L2/MALL hits included).  traffic = "
                        "PMC bytes per launch of the same build (FETCH_SIZE x2 + WRITE_SIZE, separate passes); FETCH_SIZE "
                        "counts L2->fabric requests, Infinity-Cache hits INCLUDED (MI355X_MICROARCH.md): with "
                        "hot_set_in_flight_bytes (QPs in flight x mean final free set x one column) of the order of the "
                        "256 MiB Infinity Cache, frac_from_traffic is an upper bound on the DRAM share.  About 6.3 TB/s "
                        "are achievable.  Serial single launches: roofline_serial; the reference-shaped formulation: "
                        "roofline_dense_formulation.  With lazy_handover a launch that owed a later stage has its end event "
                        "re-recorded behind that stage, host gap included (cfg4 owes none; cfg3 is timed with serial launches)"}
        issue = issue_block(sq, k_ms / conc, float(iters.sum()) / len(used), "the timed build")
        roof_serial = {"bound": "hbm", "achieved": read_serial / (iso_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": read_serial / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic_serial,
                       "kernel_ms": iso_ms, "alg_bytes_per_launch": read_serial, "same_S_and_iters": same_serial,
                       "qps": P / (iso_ms * 1e-3),
                       "issue": issue_block(sq_serial, iso_ms, float(res_serial["status"].sum()), "the serial four-per-CU launch"),
                       "note": "ONE launch at a time on one stream (library default for this batch size), kernel time by "
                               "the library's own HIP events: bound by its dependent single-wavefront chains"}
        out = {
            "metric": "QPs/sec (batched N=512 dense portfolio QP, solveQP(Q,S,x0) to KKT)",
            "value": qps, "unit": "QPs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "repeats": {"ms_per_step": [1e3 * e / args.steps for e in rep_s],
                        "ms_per_step_median": 1e3 * float(np.median(rep_s)) / args.steps,
                        "value_median": world * P * args.steps / float(np.median(rep_s)),
                        "note": "the same timed region repeated; `value` / `ms_per_step` are the first one"},
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d QPs/GPU per step, N=%d M=%d J=%d, V=X'X/T+%g*I, box [0,%g], Phase-1 vertex "
                                   "resident in HBM; %d distinct batches per GPU (seeds %d + lane*%d), one per launch lane"
                                   % (args.config, P, N, batch.M, J, cfg.delta, cfg.ub, nlanes, lanes[0].seed0, P),
                       "qps_per_gpu": P,
                       "compute_units": num_cu, "options": dict(kv.split("=", 1) for kv in args.option),
                       "parallelism": "one QP per wavefront (%d per CU in the timed region), batch sharded over %d GPU(s)"
                                      % (qpc_timed, world),
                       "formulation": "dense (reference-shaped)" if args.dense else "default (kept factor, cached products)"},
            "iters_to_kkt": {"mean": float(iters.mean()), "max": int(iters.max()), "min": int(iters.min())},
            "all_converged": ok,
            "lean_builds": {"timed_without_statistics": not args.keep_stats, "same_z_S_status_as_the_accounting_launches": lean_same,
                            "note": "the timed launches pass stats = NULL and ntrace = 0 (a production caller's call): the "
                                    "library runs the kernel builds without the byte / flop accounting; bytes, passes and "
                                    "kernel shares in this line come from one accounting launch per lane of the same "
                                    "configuration"},
            "kernels": {"wavefront_kernel_share": wave_share, "handed_over": handed_over,
                        "continued_in_big_factor_wave_kernel": big_wave, "mean_final_free_set": mean_maxk},
            "pipeline": {"mode": mode, "streams": nlanes if mode == "lanes" else 1, "batches": nlanes,
                         "batches_distinct": distinct, "probe_ms_per_step": {k: 1e3 * v for k, v in probe.items()},
                         "kernel_ms_last_timed_launch": last_kernel_ms,
                         "kernel_ms_min_max": [float(np.min(kern_ms)), float(np.max(kern_ms))],
                         "solve_call_ms_mean": float(np.mean(launch_ms)),
                         "note": "steps are independent batches (one distinct batch per lane, own seeds); mode lanes = "
                                 "issued round-robin on `streams` HIP streams (one context per lane), the drain of one "
                                 "launch overlaps the ramp-up of the next; mode serial = the same batches one after the "
                                 "other on one stream.  Both modes are probed after warm-up and the faster one is timed; "
                                 "every step solves all its QPs from (x0, S0)"},
            "collective": {"gathers": gathers_timed, "per_step": 1 if world > 1 else 0,
                           "bytes_per_rank_per_step": (lanes[0].pg.bytes if world > 1 else 0),
                           "cuda_allocations_in_timed_region": int(alloc1 - alloc0),
                           "of_which_inside_the_backend_collective_call": int(state["coll_allocs"]),
                           "note": "one all_gather_into_tensor of the packed (z, S, status) per step into a receive buffer "
                                   "allocated once; this code allocates nothing per step (the gloo backend of the "
                                   "rehearsal mode stages device tensors through temporaries of its own)"},
            "roofline": roof,
            "roofline_issue": issue,
            "roofline_serial": roof_serial,
            "roofline_dense_formulation": None if dense is None else {
                "bound": "hbm", "achieved": dense["read"] / (dense["ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": dense["read"] / (dense["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic": traffic_dense, "kernel_ms": dense["ms"], "alg_bytes_per_launch": dense["read"],
                "reference_formula_bytes": dense_bytes, "same_S_and_iters": dense["same"],
                "qps": P / (dense["ms"] * 1e-3),
                "note": "workgroup kernel with dense_gamma=1: from-scratch factorisation and a gamma pass that reads "
                        "all N columns of V like SSQP.jl:322,352 -- the HBM-bound formulation of the reference"},
            "end_to_end_solveQP": e2e,
            "setup_s": t_setup,
        }
        if not args.no_cpu and world == 1:
            res["V_host"] = batch.t["V"].cpu().numpy()
            legs = cpu_baseline(pkg, lanes[0].prob, lanes[0].S0, lanes[0].x0, res, args.cpu_seconds, ncpu)
            out["cpu_baseline"] = legs[0]
            out["cpu_baseline_lapack"] = legs[1]
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def usable_cores():
    """cores this process may actually use: the affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(pkg, prob, S0, x0, res, seconds, ncpu):
    """The oracle timed on the host cores on a bounded sample of the same workload, one QP per OpenMP thread, all
    cores of this process's affinity mask; the same runs are the parity check of that sample.  Two legs:
      port    the C restatement with its own loops for inv(cholesky(.)) and the dense products
      lapack  the same restatement with those operations done by LAPACK/BLAS (dpotrf + dpotri, dgemm, dgemv of the
              OpenBLAS that scipy bundles, one BLAS thread per QP) -- what Julia's LinearAlgebra calls at
              SSQP.jl:322-331,351-352"""
    import numpy as np
    from oracle import oracle as orc
    P = prob["q"].shape[0]
    prob = dict(prob)
    prob["V"] = res["V_host"]

    def leg(kind):
        lapack = kind == "lapack"
        if lapack and not orc.lapack_available():
            return None

        def run(n, nthreads):
            sub = [prob[k][:n] for k in "VAGqbgdu"]
            t = time.perf_counter()
            zo, So, sto, _, used = orc.solveQP_warm_batch(*sub, S0[:n], x0[:n], nthreads=nthreads, lapack=lapack)
            return time.perf_counter() - t, zo, So, sto, used

        probe_n = min(P, ncpu)
        dt_probe, *_ = run(probe_n, ncpu)                       # also warms the library / thread pool
        per_wave = max(dt_probe, 1e-3)                          # time of one "wave" of ncpu problems
        n = int(min(P, max(probe_n, ncpu * round(seconds / per_wave))))
        reps = int(max(1, min(8, round(seconds / max(per_wave * n / max(probe_n, 1), 1e-3)))))
        total = 0.0
        for _ in range(reps):
            dt, zo, So, sto, used = run(n, ncpu)
            total += dt
        scale = np.maximum(np.abs(zo).max(axis=1), 1e-300)
        rel = float((np.abs(res["z"][:n] - zo).max(axis=1) / scale).max())
        return {"value": n * reps / total, "unit": "QPs/s", "cores": int(used), "kind": "port",
                "arithmetic": "LAPACK/BLAS (scipy's OpenBLAS: dpotrf, dpotri, dgemm, dgemv; 1 BLAS thread per QP)"
                              if lapack else "hand-written loops",
                "sample": "first %d QPs of the same batch x %d repetitions, one QP per OpenMP thread, %.1f s of "
                          "wall time" % (n, reps, total),
                "parity_on_sample": {"S_bit_exact": bool(np.array_equal(res["S"][:n], So)),
                                     "iters_equal": bool(np.array_equal(res["status"][:n], sto)),
                                     "z_max_rel_err": rel}}
    return leg("port"), leg("lapack")


if __name__ == "__main__":
    main()
