#!/usr/bin/env python3
"""bench.py -- QPs/sec of the in-kernel active-set loop on batched N=512 dense portfolio QPs.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (solveQP(Q,S,x0), SSQP.jl:237-377) over one batch of
synthetic QPs already resident in HBM: BASELINE.json configs[3] ("8192 independent N=512 QPs
sharded across 8 GPUs") = 1024 QPs per GPU, cfg2-style problems (M=1, J=10, box bounds).
Weak scaling: every rank owns 1024 problems; the only collective is the final RCCL all-gather of
(z, S, status), inside the timed region, issued from ONE communication stream.  Rank 0 prints ONE JSON line.

Consecutive steps are independent batches; they are issued round-robin on `--streams` HIP streams (default 3, one
library context per lane), so the drain of one launch -- QPs need 143..262 passes, the launch ends with the slowest --
overlaps the start of the next.  `--streams 1` gives serial launches; the JSON carries both figures (`pipeline`), and
the roofline figures always come from single launches timed by their own HIP events.
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
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4.0
KERNEL_SOURCES = ["ssqp_wave.hip", "ssqp_kernels.hip", "ssqp_device.h", "ssqp_internal.h", "ssqp_api.hip"]


def kernel_source_hash():
    """sha256 over the kernel sources: PMC figures under profiles/ are only quoted for the code they were taken on"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "statusswitchingqp.jl_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nprob", type=int, default=1024, help="QPs per GPU")
    ap.add_argument("--config", default="cfg4", help="problem family (statusswitchingqp.jl_amd CONFIGS)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target CPU time of each cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--streams", type=int, default=3,
                    help="launch lanes: consecutive steps go round-robin to this many HIP streams (1: serial launches)")
    ap.add_argument("--skip-dense", action="store_true",
                    help="do not time the dense-formulation launches (profiling runs want one kernel variant)")
    ap.add_argument("--dense", action="store_true", help="time the dense (reference-shaped) formulation as the main run")
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

    cfg = pkg.CONFIGS[args.config]
    ncpu = usable_cores()
    gen_threads = max(1, ncpu // max(1, min(world, 8)))
    t0 = time.time()
    ctx = pkg.Context(local)
    if args.dense:
        ctx.set_option("dense_gamma", 1)
    # several launch lanes on one GPU: the (normally empty) hand-over launch is issued lazily, see include/ssqp_hip.h:
    # a lane's results are complete once its context was flushed (which the next solve on the lane does anyway), so
    # with more than one rank the gather of a step is issued when its lane comes round again, right after that flush
    lazy = 1 if args.streams > 1 else 0
    ctx.set_option("lazy_handover", lazy)
    # wavefronts per CU of the wavefront kernel: with several batches in flight two per SIMD (8 per CU) give more QPs/s;
    # a single 1024-QP launch fills 4 per CU exactly and runs fastest with one per SIMD (include/ssqp_hip.h)
    qpc_lanes = 8 if args.streams > 1 else 4
    ctx.set_option("wave_qp_per_cu", qpc_lanes)
    # V (the N*N*T part) is generated on the GPU, bit-identical to the host generator; the small arrays and the
    # Phase-1 vertex (x0, S0) come from the host C++ (not timed: the metric is the hot path solveQP(Q,S,x0))
    batch, prob, x0, S0 = pkg.DeviceBatch.generated(cfg, args.nprob, pkg.BASE_SEED + rank * args.nprob, ctx=ctx,
                                                    device=local, nthreads=gen_threads)
    torch.cuda.synchronize(dev)
    t_setup = time.time() - t0
    P, N, J = batch.P, batch.N, batch.J
    stream = torch.cuda.current_stream(dev)
    # Launch lanes: consecutive steps are independent batches, issued round-robin on `--streams` HIP streams (one
    # context = workspace + work counters per lane, inputs shared, outputs per lane).  A launch ends with its slowest
    # QP (143..262 passes): with a single stream the slots of the QPs that finished early idle until then; with a few
    # lanes the next launch's wavefronts take them.  Every step still solves all its QPs from (x0, S0).
    nlanes = max(1, args.streams)
    lanes = [(batch, stream)]
    for _ in range(1, nlanes):
        c2 = pkg.Context(local)
        c2.set_option("lazy_handover", lazy)
        c2.set_option("wave_qp_per_cu", qpc_lanes)
        if args.dense:
            c2.set_option("dense_gamma", 1)
        lanes.append((batch.twin(c2), torch.cuda.Stream(dev)))
    comm = torch.cuda.Stream(dev) if world > 1 else None   # the ONE stream every gather is issued from
    step_no = [0]
    owed = [False] * nlanes                 # lanes whose last solve has not been gathered yet
    gathers = [0]

    def gather(i):
        """final gather of lane i's step (RCCL over xGMI), issued from the one communication stream in launch order;
        the lane's next solve waits for it (it overwrites the buffers the gather reads)"""
        b, st = lanes[i]
        if not owed[i]:
            return
        b.ctx.flush()                       # (lazy hand-over: an owed workgroup-kernel launch goes out first)
        comm.wait_stream(st)
        with torch.cuda.stream(comm):
            pkg.dist.gather_results(b.z, b.S, b.status)
        st.wait_stream(comm)
        owed[i] = False
        gathers[0] += 1

    def step():
        i = step_no[0] % nlanes
        b, st = lanes[i]
        step_no[0] += 1
        if world > 1:
            gather(i)                       # the step this lane ran last time round
        with torch.cuda.stream(st):
            b.solve()                       # in-kernel active-set loop, asynchronous on the lane's stream
        owed[i] = world > 1

    def fence():
        if world > 1:
            first = step_no[0] % nlanes     # oldest first: launch order
            for k in range(nlanes):
                gather((first + k) % nlanes)
        for lb, lst in lanes:               # (lazy hand-over: every lane's last launch is settled)
            lb.ctx.sync(lst.cuda_stream)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(args.warmup, nlanes)):   # (every lane is warmed once)
        step()
    fence()
    step_no[0] = 0
    gathers[0] = 0
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t_start
    assert world == 1 or gathers[0] == args.steps, (gathers[0], args.steps)   # one gather per timed step, all inside
    # kernel duration of the LAST timed step: HIP events the library records on the launch stream
    last_kernel_ms = lanes[(args.steps - 1) % nlanes][0].ctx.last_kernel_ms()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if not rehearsal else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    res = batch.results()
    lanes_agree = True
    for lb, _ in lanes[1:]:                 # every lane solved the same batch: same decisions, all converged
        r2 = lb.results()
        lanes_agree = lanes_agree and bool(np.array_equal(r2["S"], res["S"]) and
                                           np.array_equal(r2["status"], res["status"]))
    ok = bool((res["status"] > 0).all()) and lanes_agree
    stats = res["stats"]
    read_bytes = int(stats["read_bytes"].sum())     # bytes this kernel's formulation has to read
    dense_bytes = int(stats["alg_bytes"].sum())     # bytes of the reference's dense formulation
    iters = res["status"].astype(np.int64)
    wave_share = float(((stats["path"] & 16) != 0).mean())
    handed_over = int(((stats["path"] & 32) != 0).sum())

    # a few more launches, each timed by its own HIP events, for the roofline figure
    def timed_launches(n):
        ms = []
        for _ in range(n):
            batch.solve()
            torch.cuda.synchronize(dev)
            ms.append(ctx.last_kernel_ms())
        return ms
    # (single launches on one stream: the kernels' own duration.  With several lanes the launches of the timed
    #  region overlap, and a launch's begin-to-end time then contains slot-sharing with its neighbours.)
    ctx.set_option("wave_qp_per_cu", 4)   # (single launches from here on: one wavefront per SIMD)
    iso = timed_launches(4)
    k_ms = float(np.mean(iso))
    # the same K steps on ONE stream, for comparison with the pipelined figure
    single = None
    if nlanes > 1:
        torch.cuda.synchronize(dev)
        ts = time.perf_counter()
        for _ in range(args.steps):
            batch.solve()
        torch.cuda.synchronize(dev)
        single = (time.perf_counter() - ts) / args.steps
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
                   "note": "ssqp_phase1_batch_dev_f64 + the loop per step, one stream; the host C++ Phase-1 "
                           "(ssqp_phase1_batch_f64) is the alternative when M + J is large"}
            if nlanes > 1 and world == 1:
                # the same with the launch lanes of the timed region: Phase-1 and loop of a step on its lane's stream
                # (each lane writes its own vertex), so one step's Phase-1 runs beside another step's loop
                ctx.set_option("wave_qp_per_cu", qpc_lanes)
                for lb, _ in lanes[1:]:
                    lb.x0, lb.S0 = lb.x0.clone(), lb.S0.clone()

                def e2e_steps(n):
                    for i in range(n):
                        lb, lst = lanes[i % nlanes]
                        with torch.cuda.stream(lst):
                            lb.phase1()
                            lb.solve()
                    for lb, lst in lanes:
                        lb.ctx.sync(lst.cuda_stream)
                    torch.cuda.synchronize(dev)
                e2e_steps(nlanes)
                ts = time.perf_counter()
                e2e_steps(args.steps)
                dtl = (time.perf_counter() - ts) / args.steps
                same = True
                for lb, _ in lanes:
                    r3 = lb.results()
                    same = same and bool(np.array_equal(r3["S"], res["S"]) and np.array_equal(r3["status"], res["status"]))
                e2e["lanes"] = {"streams": nlanes, "qps": P / dtl, "ms_per_step": 1e3 * dtl, "same_S_and_iters": same}
                ctx.set_option("wave_qp_per_cu", 4)
        except Exception as exc:   # (e.g. M + J too large for the GPU Phase-1)
            e2e = {"error": str(exc)}
    # the same problem through the workgroup kernel with the gamma pass reading EVERY column of V, as the reference's
    # dense V[B,F]*alpha + V[B,B]*zB does (SSQP.jl:352): the HBM-bound formulation, timed beside the default one
    dense = None
    if not args.skip_dense and not args.dense:
        with ctx.options(dense_gamma=1):
            dense_ms = float(np.mean(timed_launches(2)))
            res_dense = batch.results()
        dense = {"ms": dense_ms, "read": int(res_dense["stats"]["read_bytes"].sum()),
                 "same": bool(np.array_equal(res_dense["S"], res["S"]) and
                              np.array_equal(res_dense["status"], res["status"]))}

    # PMC figures are quoted only when they were collected on exactly this kernel source and workload
    pmc = None
    if os.path.exists(args.pmc_json):
        try:
            with open(args.pmc_json) as f:
                pj = json.load(f)
            if (pj.get("config") == args.config and pj.get("nprob") == args.nprob and
                    pj.get("kernel_source_sha256") == kernel_source_hash()):
                pmc = pj
        except Exception:
            pmc = None
    key = "dense_formulation" if args.dense else "default_formulation"
    traffic = pmc[key].get("hbm_bytes_per_launch") if pmc and key in pmc else None
    traffic_dense = pmc["dense_formulation"].get("hbm_bytes_per_launch") if pmc and "dense_formulation" in pmc else None
    sq = pmc[key].get("sq") if pmc and key in pmc else None
    traffic8 = (pmc["eight_per_cu_build"].get("hbm_bytes_per_launch")
                if pmc and "eight_per_cu_build" in pmc and qpc_lanes == 8 and not args.dense else None)

    out = None
    if rank == 0:
        qps = world * P * args.steps / elapsed
        achieved = read_bytes / (k_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "ssqp_wave_kernel (+ ssqp_solve_kernel for handed-over QPs)" if wave_share > 0 else "ssqp_solve_kernel",
                "kernel_ms": k_ms, "alg_bytes_per_launch": read_bytes,
                "achieved_from_traffic": None if traffic is None else traffic / (k_ms * 1e-3) / 1e9,
                "frac_from_traffic": None if traffic is None else traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                # the timed region itself (launch lanes overlap, so per-launch durations do not add up): bytes of all
                # its launches / its wall time
                "achieved_timed_region": world * read_bytes * args.steps / elapsed / 1e9,
                "frac_timed_region": read_bytes * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,
                "traffic_timed_region_build": traffic8,
                "note": "achieved = bytes this formulation reads (counted in-kernel, L2/MALL hits included) / kernel time of "
                        "ONE serial launch (four QPs per CU: bound by its dependent single-wavefront chains, see "
                        "roofline_issue); achieved_from_traffic = PMC bytes (L2 misses: FETCH_SIZE x2 + WRITE_SIZE) / the same "
                        "time.  *_timed_region = the bytes of all launches of the timed region / its wall time: with launch "
                        "lanes (eight QPs per CU, overlapping launches) the same formulation moves its bytes at that rate, "
                        "which is where HBM starts to bound it (about 6.3 TB/s are achievable).  The reference-shaped "
                        "formulation of the same path is roofline_dense_formulation"}
        issue = None
        if sq:
            issue = {"bound": "valu-issue", "achieved": sq["SQ_INSTS_VALU"] / (k_ms * 1e-3), "peak": VALU_ISSUE_PEAK,
                     "unit": "wave-instructions/s", "frac": sq["SQ_INSTS_VALU"] / (k_ms * 1e-3) / VALU_ISSUE_PEAK,
                     "issue_active_frac": sq["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
                     "wait_frac": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
                     "valu_per_pass": sq["SQ_INSTS_VALU"] / float(iters.sum()),
                     "salu_per_pass": sq["SQ_INSTS_SALU"] / float(iters.sum()),
                     "achieved_timed_region": sq["SQ_INSTS_VALU"] / (elapsed / args.steps),
                     "frac_timed_region": sq["SQ_INSTS_VALU"] / (elapsed / args.steps) / VALU_ISSUE_PEAK,
                     "note": "SQ counters per launch from profiles/ (same kernel source hash; serial four-per-CU launch) / "
                             "this run's kernel time; *_timed_region: / the wall time per step of the timed region (launch "
                             "lanes, eight per CU: same instruction stream per QP up to the parked row slot); peak = 256 "
                             "CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 f64 VALU instruction"}
        out = {
            "metric": "QPs/sec (batched N=512 dense portfolio QP, solveQP(Q,S,x0) to KKT)",
            "value": qps, "unit": "QPs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d QPs/GPU, N=%d M=%d J=%d, V=X'X/T+%g*I, box [0,%g], Phase-1 vertex "
                                   "resident in HBM" % (args.config, P, N, batch.M, J, cfg.delta, cfg.ub),
                       "qps_per_gpu": P,
                       "parallelism": "one QP per wavefront (%d per CU in the timed region), batch sharded over %d GPU(s)"
                                      % (qpc_lanes, world),
                       "formulation": "dense (reference-shaped)" if args.dense else "default (kept factor, cached products)"},
            "iters_to_kkt": {"mean": float(iters.mean()), "max": int(iters.max()), "min": int(iters.min())},
            "all_converged": ok,
            "kernels": {"wavefront_kernel_share": wave_share, "handed_over_to_workgroup_kernel": handed_over},
            "pipeline": {"streams": nlanes, "lanes_agree": lanes_agree,
                         "kernel_ms_last_timed_launch": last_kernel_ms,
                         "single_stream_ms_per_step": None if single is None else 1e3 * single,
                         "single_stream_qps": None if single is None else P / single,
                         "wave_qp_per_cu": qpc_lanes,
                         "note": "steps are independent batches issued round-robin on `streams` HIP streams (one "
                                 "context per lane, shared inputs, per-lane outputs): the drain of one launch overlaps "
                                 "the ramp-up of the next; every step solves all its QPs from (x0, S0)"},
            "roofline": roof,
            "roofline_issue": issue,
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
            legs = cpu_baseline(pkg, prob, S0, x0, res, args.cpu_seconds, ncpu)
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
