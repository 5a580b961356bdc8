// ssqp_api.hip -- the C ABI of libssqp_hip.so (include/ssqp_hip.h): context,
// workspaces, launches.  The solver itself is ssqp_kernels.hip; there is no
// CPU fallback anywhere in this library: without a HIP device
// ssqp_ctx_create fails with SSQP_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "ssqp_hip.h"
#include "ssqp_internal.h"

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct ssqp_ctx {
    int device = 0;
    int numCU = 0;
    hipStream_t stream = nullptr;
    // begin / end events of the solve kernels of the last EV_RING launches (ssqp_last_kernel_ms, ssqp_recent_kernel_ms)
    static constexpr int EV_RING = 16;
    hipEvent_t evB[EV_RING] = {}, evE[EV_RING] = {};
    unsigned long long nLaunch = 0;   // launches so far; launch k uses slot k % EV_RING
    std::string err;
    // options (ssqp_ctx_set_option): algorithm switches are per context, never read from the environment
    int optWgPerCU = 0;      // 0 = automatic
    int optDenseGamma = 0;   // 1: dense (reference-shaped) formulation -- roofline measurements
    int optIncremental = 1;  // 0: refactor V[F,F] from scratch in every pass
    int optWaveKernel = 1;   // 0: never use the wavefront-per-QP kernel; 2: start in its big-factor build
    int optWaveQPC = 0;      // QPs (wavefronts) per CU of the wavefront kernel: 0 = by batch size, 1..4, 8
    int optLazyHandover = 0; // 1: the hand-over launch is deferred to ssqp_sync / the next call and skipped when empty
    int optPinHost = 0;      // 1: page-lock the caller's V array (kept registered until another array comes)
    const void *pinnedPtr = nullptr;
    size_t pinnedBytes = 0;
    // grow-only device workspaces
    DevBuf Ct, rhs, queue, gscratch, fbList, fbList2, fbIter, wscratch, wscratchBig, p1ws, p1wsInt, p1queue, p1list, fullX0, fullSt;
    int optPhase1Wave = 1;   // 0: Phase-1 by the workgroup kernel only
    // staging buffers of the host-pointer entry points
    DevBuf hV, hA, hG, hq, hb, hg, hd, hu, hS, hx0, hz, hstatus, hdetail, hstats, hlam, hgam;
    // lazy hand-over: the launch the wavefront kernel may still owe (its hand-over count lands in pinned memory)
    unsigned int *hostCount = nullptr;   // pinned
    hipEvent_t evCount = nullptr;
    bool pending = false;
    ssqp::SolveParams pendP;          // parameters of the workgroup-kernel stage
    ssqp::SolveParams pendBig;        // ... of the big-factor wavefront stage in front of it (pendBigGrid > 0)
    int pendGrid = 0, pendWg = 0, pendBigGrid = 0, pendSlot = 0;
    size_t pendLds = 0;
    hipStream_t pendStream = nullptr;
    hipStream_t owedStream = nullptr;    // stream an owed hand-over launch went out on (ordered before the next call's resets)
    hipEvent_t evOwed = nullptr;
    // launch lanes of the host-buffer batch entry: child contexts (own stream, workspaces, work counters) so that
    // the solve of one chunk overlaps the upload of the next and the solves of neighbouring chunks
    std::vector<ssqp_ctx *> lanes;
    hipEvent_t evCopy[4] = {nullptr, nullptr, nullptr, nullptr};
};

// a batch resident in HBM (ssqp_problem_upload): solved any number of times without moving V again
struct ssqp_problem {
    ssqp_ctx *ctx = nullptr;
    int nprob = 0, N = 0, M = 0, J = 0;
    DevBuf V, A, G, q, b, g, d, u, S, x0, z, status, detail, stats;
};

namespace {

bool hip_ok(ssqp_ctx *c, hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    if (c) c->err = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

bool ensure(ssqp_ctx *c, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 8;
    if (b.bytes >= bytes) return true;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
    if (!hip_ok(c, hipMalloc(&b.p, bytes), "hipMalloc")) return false;
    b.bytes = bytes;
    return true;
}

void release(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

int check_dims(ssqp_ctx *c, int nprob, int N, int M, int J) {
    if (!c) return SSQP_ERR_ARG;
    if (nprob < 0 || N <= 0 || M < 0 || J < 0) {
        c->err = "bad dimensions";
        return SSQP_ERR_ARG;
    }
    if (N > ssqp::MAXN || N + J + 2 > 32767 || M + J + 2 > 32767) {
        c->err = "N above the in-kernel limit (2048)";
        return SSQP_ERR_UNSUPPORTED;
    }
    if (ssqp::lds_fixed_bytes(N, M, J) + 1024 > ssqp::LDS_BYTES) {
        c->err = "N/M/J vectors do not fit in LDS";
        return SSQP_ERR_UNSUPPORTED;
    }
    return SSQP_OK;
}

}  // namespace

extern "C" {

int ssqp_ctx_create(int device, ssqp_ctx **out) {
    if (!out) return SSQP_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return SSQP_ERR_NO_DEVICE;
    ssqp_ctx *c = new (std::nothrow) ssqp_ctx();
    if (!c) return SSQP_ERR_ALLOC;
    c->device = device;
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
        delete c;
        return SSQP_ERR_NO_DEVICE;
    }
    c->numCU = prop.multiProcessorCount;
    bool evOk = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; k < ssqp_ctx::EV_RING && evOk; ++k)
        evOk = hipEventCreate(&c->evB[k]) == hipSuccess && hipEventCreate(&c->evE[k]) == hipSuccess;
    if (!evOk) {
        (void)ssqp_ctx_destroy(c);
        return SSQP_ERR_HIP;
    }
    *out = c;
    return SSQP_OK;
}

int ssqp_ctx_destroy(ssqp_ctx *c) {
    if (!c) return SSQP_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->pinnedPtr) (void)hipHostUnregister(const_cast<void *>(c->pinnedPtr));
    c->pinnedPtr = nullptr;
    if (c->hostCount) (void)hipHostFree(c->hostCount);
    c->hostCount = nullptr;
    if (c->evCount) (void)hipEventDestroy(c->evCount);
    c->evCount = nullptr;
    if (c->evOwed) (void)hipEventDestroy(c->evOwed);
    c->evOwed = nullptr;
    for (ssqp_ctx *l : c->lanes) (void)ssqp_ctx_destroy(l);
    c->lanes.clear();
    for (hipEvent_t &e : c->evCopy)
        if (e) (void)hipEventDestroy(e), e = nullptr;
    for (DevBuf *b : {&c->Ct, &c->rhs, &c->queue, &c->gscratch, &c->fbList, &c->fbList2, &c->fbIter, &c->wscratch, &c->wscratchBig, &c->p1ws, &c->p1wsInt, &c->p1queue, &c->p1list, &c->fullX0, &c->fullSt, &c->hV, &c->hA, &c->hG, &c->hq, &c->hb, &c->hg,
                      &c->hd, &c->hu, &c->hS, &c->hx0, &c->hz, &c->hstatus, &c->hdetail, &c->hstats, &c->hlam, &c->hgam})
        release(*b);
    for (int k = 0; k < ssqp_ctx::EV_RING; ++k) {
        if (c->evB[k]) (void)hipEventDestroy(c->evB[k]);
        if (c->evE[k]) (void)hipEventDestroy(c->evE[k]);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SSQP_OK;
}

const char *ssqp_last_error(const ssqp_ctx *c) { return c ? c->err.c_str() : "null context"; }

static int *option_slot(ssqp_ctx *c, const char *name) {
    if (!c || !name) return nullptr;
    if (!std::strcmp(name, "wg_per_cu")) return &c->optWgPerCU;
    if (!std::strcmp(name, "dense_gamma")) return &c->optDenseGamma;
    if (!std::strcmp(name, "incremental")) return &c->optIncremental;
    if (!std::strcmp(name, "wave_kernel")) return &c->optWaveKernel;
    if (!std::strcmp(name, "wave_qp_per_cu")) return &c->optWaveQPC;
    if (!std::strcmp(name, "pin_host_buffers")) return &c->optPinHost;
    if (!std::strcmp(name, "lazy_handover")) return &c->optLazyHandover;
    if (!std::strcmp(name, "phase1_wave")) return &c->optPhase1Wave;
    return nullptr;
}
int ssqp_ctx_set_option(ssqp_ctx *c, const char *name, int value) {
    int *slot = option_slot(c, name);
    if (!slot) {
        if (c) c->err = std::string("unknown option ") + (name ? name : "(null)");
        return SSQP_ERR_ARG;
    }
    if ((slot == &c->optWgPerCU && (value < 0 || value > ssqp::MAX_WG_PER_CU)) ||
        (slot == &c->optWaveQPC && (value < 0 || value > 8)) ||
        (slot == &c->optWaveKernel && (value < 0 || value > 2)) ||
        ((slot == &c->optDenseGamma || slot == &c->optIncremental || slot == &c->optPinHost || slot == &c->optLazyHandover ||
          slot == &c->optPhase1Wave) &&
         (value != 0 && value != 1))) {
        c->err = std::string("option value out of range: ") + name;
        return SSQP_ERR_ARG;
    }
    *slot = value;
    if (slot == &c->optPinHost && value == 0 && c->pinnedPtr) {  // the registration ends with the option
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (hipHostUnregister(const_cast<void *>(c->pinnedPtr)) != hipSuccess) (void)hipGetLastError();
        c->pinnedPtr = nullptr;
        c->pinnedBytes = 0;
    }
    return SSQP_OK;
}
int ssqp_ctx_get_option(ssqp_ctx *c, const char *name, int *value) {
    int *slot = option_slot(c, name);
    if (!slot || !value) return SSQP_ERR_ARG;
    *value = *slot;
    return SSQP_OK;
}

// lazy hand-over: wait for the wavefront kernel of the last call, and launch the workgroup kernel on its hand-over
// list only when that list is not empty (on the stream of that call, so later work on it stays ordered)
static int finish_pending(ssqp_ctx *c) {
    if (!c->pending) return SSQP_OK;
    c->pending = false;
    if (!hip_ok(c, hipEventSynchronize(c->evCount), "hipEventSynchronize")) return SSQP_ERR_HIP;
    if (*c->hostCount == 0) return SSQP_OK;
    if (c->pendBigGrid > 0 &&
        !hip_ok(c, ssqp::launch_solve_wave(c->pendBig, c->pendBigGrid, 2, c->pendStream), "big-factor wave launch"))
        return SSQP_ERR_HIP;
    if (!hip_ok(c, ssqp::launch_solve(c->pendP, c->pendGrid, c->pendLds, c->pendWg, c->pendStream), "solve launch"))
        return SSQP_ERR_HIP;
    if (!hip_ok(c, hipEventRecord(c->evE[c->pendSlot], c->pendStream), "hipEventRecord")) return SSQP_ERR_HIP;
    c->owedStream = c->pendStream;
    return SSQP_OK;
}

// The owed launch went out on the stream of the call that owed it.  A later call on ANOTHER stream resets the shared
// work counters and (through its caller) the in/out buffers: it has to queue behind that launch.
static int order_after_owed(ssqp_ctx *c, hipStream_t s) {
    if (!c->owedStream) return SSQP_OK;
    hipStream_t o = c->owedStream;
    c->owedStream = nullptr;
    if (o == s) return SSQP_OK;
    if (!c->evOwed && !hip_ok(c, hipEventCreateWithFlags(&c->evOwed, hipEventDisableTiming), "hipEventCreate")) return SSQP_ERR_HIP;
    if (!hip_ok(c, hipEventRecord(c->evOwed, o), "hipEventRecord") || !hip_ok(c, hipStreamWaitEvent(s, c->evOwed, 0), "hipStreamWaitEvent"))
        return SSQP_ERR_HIP;
    return SSQP_OK;
}

int ssqp_flush(ssqp_ctx *c) {
    if (!c) return SSQP_ERR_ARG;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return SSQP_ERR_HIP;
    return finish_pending(c);
}

int ssqp_flush_to(ssqp_ctx *c, void *stream) {
    if (!c) return SSQP_ERR_ARG;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return SSQP_ERR_HIP;
    const int rc = finish_pending(c);
    return rc != SSQP_OK ? rc : order_after_owed(c, (hipStream_t)stream);
}

int ssqp_sync(ssqp_ctx *c, void *stream) {
    if (!c) return SSQP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;  // NULL is HIP's default stream, as everywhere in HIP
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return SSQP_ERR_HIP;
    const int rc = finish_pending(c);
    if (rc != SSQP_OK) return rc;
    return hip_ok(c, hipStreamSynchronize(s), "hipStreamSynchronize") ? SSQP_OK : SSQP_ERR_HIP;
}

int ssqp_recent_kernel_ms(ssqp_ctx *c, int back, float *ms) {
    if (!c || !ms || back < 0 || back >= ssqp_ctx::EV_RING) return SSQP_ERR_ARG;
    if (c->nLaunch <= (unsigned long long)back) {
        c->err = "no such launch on this context";
        return SSQP_ERR_ARG;
    }
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return SSQP_ERR_HIP;
    {
        const int rcp = finish_pending(c);
        if (rcp != SSQP_OK) return rcp;
    }
    const int slot = (int)((c->nLaunch - 1 - (unsigned long long)back) % ssqp_ctx::EV_RING);
    if (!hip_ok(c, hipEventSynchronize(c->evE[slot]), "hipEventSynchronize")) return SSQP_ERR_HIP;
    return hip_ok(c, hipEventElapsedTime(ms, c->evB[slot], c->evE[slot]), "hipEventElapsedTime") ? SSQP_OK : SSQP_ERR_HIP;
}

int ssqp_last_kernel_ms(ssqp_ctx *c, float *ms) { return ssqp_recent_kernel_ms(c, 0, ms); }

}  // extern "C"

// the device-buffer solve; `settingsLP` non-null = solveQP(Q) in one launch: Phase-1 runs in front of the loop inside the
// first stage's kernel (dx0 is then null: the vertex lives in a workspace of the context)
static int solve_dev_impl(ssqp_ctx *c, int nprob, int N, int M, int J, const double *dV, const double *dA,
                          const double *dG, const double *dq, const double *db, const double *dg,
                          const double *dd, const double *du, const ssqp_batch_strides *strides,
                          int32_t *dS, const double *dx0, double *dz, const ssqp_settings *settings,
                          int64_t *dstatus, int32_t *ddetail, ssqp_stats *dstats, ssqp_trace *dtrace,
                          int ntrace, double *dlambda, double *dgamma, void *stream, const ssqp_settings *settingsLP) {
    const bool full = settingsLP != nullptr;
    ssqp_batch_strides st0;
    st0.V = (size_t)N * N; st0.A = (size_t)M * N; st0.G = (size_t)J * N; st0.q = N; st0.b = M; st0.g = J; st0.d = N; st0.u = N;
    const ssqp_batch_strides *sd = strides ? strides : &st0;
    int rc = check_dims(c, nprob, N, M, J);
    if (rc != SSQP_OK) return rc;
    if (!dV || !dq || !dd || !du || !dS || (!dx0 && !full) || !dz || !dstatus || (M > 0 && (!dA || !db)) ||
        (J > 0 && (!dG || !dg))) {
        c->err = "null pointer";
        return SSQP_ERR_ARG;
    }
    if (nprob == 0) return SSQP_OK;
    ssqp_settings def;
    ssqp_default_settings(&def);
    const ssqp_settings *st = settings ? settings : &def;
    if (full) {
        if (settingsLP->rule != 0) {
            c->err = "only rule = :Dantzig is implemented for Phase-1";
            return SSQP_ERR_UNSUPPORTED;
        }
        const bool packed = sd->A == (size_t)M * N && sd->G == (size_t)J * N && sd->b == (size_t)M && sd->g == (size_t)J &&
                            sd->d == (size_t)N && sd->u == (size_t)N;
        if (!packed || !ssqp::phase1_wave_applies(N, M, J) || !ssqp::wave_kernel_applies(N, M, J) || !c->optWaveKernel ||
            c->optWaveKernel == 2 || !c->optIncremental || c->optDenseGamma || !c->optPhase1Wave) {
            c->err = "single-launch solveQP(Q) takes per-problem A, G, b, g, d, u with N even <= 512, 1 <= M + J <= 11, "
                     "N + J + M + J <= 576, default options: use ssqp_phase1_batch_dev_f64 + ssqp_solve_batch_dev_f64";
            return SSQP_ERR_UNSUPPORTED;
        }
    }
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return SSQP_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;  // NULL is HIP's default stream (what torch uses unless told otherwise)
    {
        int rcp = finish_pending(c);  // (lazy hand-over of the previous call on this context)
        if (rcp == SSQP_OK) rcp = order_after_owed(c, s);
        if (rcp != SSQP_OK) return rcp;
    }

    const int MJ = M + J;
    // the multipliers are written for status > 0 only: every call starts them from zero (include/ssqp_hip.h)
    if (dlambda && MJ > 0 && !hip_ok(c, hipMemsetAsync(dlambda, 0, (size_t)nprob * MJ * 8, s), "hipMemsetAsync")) return SSQP_ERR_HIP;
    if (dgamma && !hip_ok(c, hipMemsetAsync(dgamma, 0, (size_t)nprob * N * 8, s), "hipMemsetAsync")) return SSQP_ERR_HIP;
    // workgroups per CU: LDS is the limit.  3 when the N-vectors leave a useful arena in 1/3 of the 160 KiB,
    // else 2, else 1 (SSQP_WG_PER_CU overrides for experiments).  A pass whose factor does not fit the LDS
    // arena runs on the per-workgroup global arena instead, so any choice is correct.
    const int fixed = ssqp::lds_fixed_bytes(N, M, J);
    int wgPerCU = 1;
    for (int w = 2; w >= 1; --w) {  // 3 per CU costs register spills at 168 VGPRs and measured slower
        const int per = (ssqp::LDS_BYTES / w) / 1024 * 1024;
        if (per - fixed >= 32 * 1024 || w == 1) {
            wgPerCU = w;
            break;
        }
    }
    if (c->optWgPerCU >= 1) wgPerCU = c->optWgPerCU;
    if (N > 512 && (N & 1) == 0) wgPerCU = 1;  // the wide-accumulator kernels are built for one workgroup per CU
    const int ldsPerWG = (ssqp::LDS_BYTES / wgPerCU) / 1024 * 1024;
    int grid = c->numCU * wgPerCU;
    if (grid > nprob) grid = nprob;
    if (grid < 1) grid = 1;
    const size_t gstride = ssqp::global_arena_doubles(N, M, J);
    // the wavefront-per-QP kernel takes the shapes it is built for (N even <= 512, M + J <= 11) in the default
    // formulation.  A QP whose free set outgrows the factor of the build it started in is handed over: builds 0 / 1
    // (up to 92 / 127 rows) -> the big-factor build (four row slots, up to WAVE_BIG_ROWS rows) -> the workgroup kernel
    const bool useWave = c->optWaveKernel && c->optIncremental && !c->optDenseGamma && ssqp::wave_kernel_applies(N, M, J);
    const bool bigFirst = useWave && c->optWaveKernel == 2;   // start in the big-factor build (option)
    const bool bigStage = useWave && (bigFirst || N > 92);    // (a smaller N can never outgrow build 0 / 1 ... nearly)
    int waveGrid = 0, waveRC = 0, waveLds = 0, waveWps = 1, bigGrid = 0;
    size_t wstride = 0, wstrideBig = 0;
    if (useWave) {
        // 4 per CU (one wavefront per SIMD, 512 registers, everything in LDS) is the faster kernel per QP; 8 per CU (two per
        // SIMD, 256 registers, rows >= 64 of the factor and -- between the passes that use it -- the second row slot in
        // global scratch) hides each wavefront's waits behind another one: the better choice when more QPs are in flight
        // than 4 per CU -- a batch above 4 * numCU QPs, or several contexts busy on different streams (the caller says
        // so with the option)
        // (the single-launch solveQP(Q) is the four-per-CU build: Phase-1 keeps the LP's columns in 512 registers)
        const int qpc = full ? 4 : (c->optWaveQPC > 0 ? c->optWaveQPC : (nprob > 4 * c->numCU ? 8 : 4));
        if (qpc > 4) {  // two wavefronts per SIMD: 256 registers, rows >= 64 of the factor in global scratch
            waveWps = 2;
            waveRC = N < 127 ? N : 127;
            waveLds = ssqp::wave_lds_bytes(0);
        } else {
            const int perWave = (ssqp::LDS_BYTES / qpc) / 256 * 256;
            waveRC = 127;
            while (waveRC > 8 && ssqp::wave_lds_bytes(waveRC) > perWave) --waveRC;
            if (waveRC > N) waveRC = N;
            waveLds = ssqp::wave_lds_bytes(waveRC);
        }
        waveGrid = c->numCU * qpc;
        if (waveGrid > nprob) waveGrid = nprob;
        wstride = ssqp::wave_scratch_doubles(waveWps == 2 ? 1 : 0);
        if (bigStage) {  // one wavefront per SIMD (its four row slots take the whole register file)
            bigGrid = c->numCU * 4;
            if (bigGrid > nprob) bigGrid = nprob;
            wstrideBig = ssqp::wave_scratch_doubles(2);
        }
    }
    if (!ensure(c, c->Ct, (size_t)nprob * MJ * N * 8) || !ensure(c, c->rhs, (size_t)nprob * MJ * 8) ||
        !ensure(c, c->queue, 64) || !ensure(c, c->gscratch, (size_t)grid * gstride * 8))
        return SSQP_ERR_ALLOC;
    if (full && (!ensure(c, c->fullX0, (size_t)nprob * N * 8) || !ensure(c, c->fullSt, (size_t)nprob * 4) ||
                 !ensure(c, c->p1queue, 64) || !ensure(c, c->p1list, (size_t)nprob * 4) ||
                 !ensure(c, c->p1ws, (size_t)nprob * ssqp::phase1_ws_doubles(N, M, J) * 8) ||
                 !ensure(c, c->p1wsInt, (size_t)nprob * ssqp::phase1_ws_ints(N, M, J) * 4)))
        return SSQP_ERR_ALLOC;
    if (full) dx0 = (const double *)c->fullX0.p;
    if (useWave && (!ensure(c, c->fbList, (size_t)nprob * 4) || !ensure(c, c->fbList2, (size_t)nprob * 4) ||
                    !ensure(c, c->fbIter, (size_t)nprob * 8) || !ensure(c, c->wscratch, (size_t)waveGrid * wstride * 8) ||
                    (bigStage && !ensure(c, c->wscratchBig, (size_t)bigGrid * wstrideBig * 8))))
        return SSQP_ERR_ALLOC;

    ssqp::SolveParams P;
    P.nprob = nprob; P.N = N; P.M = M; P.J = J; P.MJ = MJ;
    P.V = dV;
    P.Ct = (const double *)c->Ct.p;
    P.rhs = (const double *)c->rhs.p;
    P.q = dq; P.d = dd; P.u = du; P.x0 = dx0;
    const bool sharedC = (sd->A == 0 || M == 0) && (sd->G == 0 || J == 0) && MJ > 0;
    const bool sharedR = (sd->b == 0 || M == 0) && (sd->g == 0 || J == 0) && MJ > 0;
    P.sV = sd->V; P.sq = sd->q; P.sd = sd->d; P.su = sd->u;
    P.sCt = sharedC ? 0 : (size_t)MJ * N;
    P.sRhs = sharedR ? 0 : (size_t)MJ;
    P.S = dS; P.z = dz; P.status = dstatus; P.detail = ddetail; P.stats = dstats;
    P.trace = (ntrace > 0) ? dtrace : nullptr;
    P.ntrace = (dtrace && ntrace > 0) ? ntrace : 0;
    P.lamOut = (MJ > 0) ? dlambda : nullptr;
    P.gamOut = dgamma;
    P.maxIter = st->maxIter; P.tol = st->tol; P.tolG = st->tolG;
    P.queue = (unsigned int *)c->queue.p;
    P.gscratch = (double *)c->gscratch.p;
    P.gscratchStride = gstride;
    P.denseGamma = c->optDenseGamma;
    P.incremental = c->optDenseGamma ? 0 : c->optIncremental;  // the dense run is the from-scratch, reference-shaped pass
    // queue words: work counters [0] first wavefront stage, [1] workgroup kernel, [4] big-factor stage;
    //              hand-over counts [2] out of the first stage, [3] out of the big-factor stage
    unsigned int *qw = (unsigned int *)c->queue.p;
    P.fbCount = qw + 2;
    P.fbList = (int *)c->fbList.p;
    P.fbIter = (long long *)c->fbIter.p;
    P.resume = 0;
    P.resumeCount = nullptr;
    P.resumeList = nullptr;
    P.wscratch = (double *)c->wscratch.p;
    P.wscratchStride = wstride;
    P.waveLdsBytes = waveLds;
    P.waveRC = waveRC;
    P.arenaCap = ((ldsPerWG - fixed - 64) / 16) * 2;
    if (P.arenaCap < 0) P.arenaCap = 0;
    const ssqp::LdsLayout lay = ssqp::lds_layout(N, M, J, P.arenaCap);
    if (lay.total_bytes > ssqp::LDS_BYTES) {
        c->err = "internal: LDS layout overflow";
        return SSQP_ERR_UNSUPPORTED;
    }

    if (!hip_ok(c, hipMemsetAsync(c->queue.p, 0, 64, s), "hipMemsetAsync")) return SSQP_ERR_HIP;
    if (full && !hip_ok(c, hipMemsetAsync(c->p1queue.p, 0, 64, s), "hipMemsetAsync")) return SSQP_ERR_HIP;
    ssqp::launch_prep(sharedC ? 1 : nprob, sharedR ? 1 : nprob, N, M, J, dA, dG, db, dg, sd->A, sd->G, sd->b, sd->g,
                      (double *)c->Ct.p, (double *)c->rhs.p, s);
    if (!hip_ok(c, hipGetLastError(), "prep launch")) return SSQP_ERR_HIP;
    const int slot = (int)(c->nLaunch % ssqp_ctx::EV_RING);
    c->nLaunch += 1;
    if (!hip_ok(c, hipEventRecord(c->evB[slot], s), "hipEventRecord")) return SSQP_ERR_HIP;
    if (useWave) {
        // the big-factor stage: problems from the first stage's hand-over list (or all of them when it goes first)
        ssqp::SolveParams B = P;
        B.queue = qw + 4;
        B.wscratch = (double *)c->wscratchBig.p;
        B.wscratchStride = wstrideBig;
        B.waveLdsBytes = ssqp::wave_lds_bytes_big();
        B.waveRC = N < ssqp::WAVE_BIG_ROWS ? N : ssqp::WAVE_BIG_ROWS;
        B.resume = bigFirst ? 0 : 1;
        B.resumeCount = qw + 2;
        B.resumeList = (const int *)c->fbList.p;
        B.fbCount = qw + 3;
        B.fbList = (int *)c->fbList2.p;
        // the workgroup kernel: what the last wavefront stage could not finish
        ssqp::SolveParams W = P;
        W.queue = qw + 1;
        W.resume = 1;  // (a grid that finds the hand-over list empty exits at once)
        if (bigStage) {
            W.fbCount = qw + 3;
            W.fbList = (int *)c->fbList2.p;
        }
        if (full) {
            // Phase-1 + loop in one kernel; what its Phase-1 does not take (free variables) goes through the workgroup
            // Phase-1 kernel, which appends the feasible ones to the first stage's hand-over list at pass 0
            if (!hip_ok(c, ssqp::launch_solve_full(P, waveGrid, dA, dG, db, dg, settingsLP->tol, (double *)c->fullX0.p,
                                                   (int32_t *)c->fullSt.p, (unsigned int *)c->p1queue.p, (int *)c->p1list.p, s),
                        "single-launch solve")) return SSQP_ERR_HIP;
            ssqp::Phase1Handover ho{P.fbCount, P.fbList, P.fbIter, dz, dstatus, ddetail, dstats};
            if (!hip_ok(c, ssqp::launch_phase1(nprob, N, M, J, dA, dG, db, dg, dd, du, settingsLP->tol, (double *)c->fullX0.p, dS,
                                               (int32_t *)c->fullSt.p, (double *)c->p1ws.p, ssqp::phase1_ws_doubles(N, M, J),
                                               (int *)c->p1wsInt.p, ssqp::phase1_ws_ints(N, M, J),
                                               (const unsigned int *)c->p1queue.p, (const int *)c->p1list.p, 64, &ho, s),
                        "phase-1 launch")) return SSQP_ERR_HIP;
        } else if (bigFirst) {
            if (!hip_ok(c, ssqp::launch_solve_wave(B, bigGrid, 2, s), "big-factor wave launch")) return SSQP_ERR_HIP;
        } else {
            if (!hip_ok(c, ssqp::launch_solve_wave(P, waveGrid, waveWps == 2 ? 1 : 0, s), "wave solve launch")) return SSQP_ERR_HIP;
        }
        if (c->optLazyHandover) {
            // The later stages need CUs with free LDS and register files: behind a launch of another context they would
            // wait for that even when there is nothing to do.  Lazy mode: the hand-over count of the first stage comes to
            // pinned host memory and the later stages are issued by ssqp_sync / the next call only if it is not zero.
            if (!c->hostCount && !hip_ok(c, hipHostMalloc((void **)&c->hostCount, 64, hipHostMallocDefault), "hipHostMalloc"))
                return SSQP_ERR_ALLOC;
            if (!c->evCount && !hip_ok(c, hipEventCreateWithFlags(&c->evCount, hipEventDisableTiming), "hipEventCreate"))
                return SSQP_ERR_HIP;
            if (!hip_ok(c, hipEventRecord(c->evE[slot], s), "hipEventRecord")) return SSQP_ERR_HIP;
            if (!hip_ok(c, hipMemcpyAsync(c->hostCount, bigFirst ? B.fbCount : P.fbCount, 4, hipMemcpyDeviceToHost, s), "D2H") ||
                !hip_ok(c, hipEventRecord(c->evCount, s), "hipEventRecord"))
                return SSQP_ERR_HIP;
            c->pendP = W;
            c->pendBig = B;
            c->pendBigGrid = (bigStage && !bigFirst) ? bigGrid : 0;
            c->pendGrid = grid;
            c->pendWg = wgPerCU;
            c->pendLds = (size_t)lay.total_bytes;
            c->pendStream = s;
            c->pendSlot = slot;
            c->pending = true;
            return SSQP_OK;
        }
        if (bigStage && !bigFirst &&
            !hip_ok(c, ssqp::launch_solve_wave(B, bigGrid, 2, s), "big-factor wave launch"))
            return SSQP_ERR_HIP;
        P = W;
    }
    if (!hip_ok(c, ssqp::launch_solve(P, grid, (size_t)lay.total_bytes, wgPerCU, s), "solve launch")) return SSQP_ERR_HIP;
    if (!hip_ok(c, hipEventRecord(c->evE[slot], s), "hipEventRecord")) return SSQP_ERR_HIP;
    return SSQP_OK;
}

extern "C" {

int ssqp_solve_batch_strided_dev_f64(ssqp_ctx *c, int nprob, int N, int M, int J, const double *dV, const double *dA,
                                     const double *dG, const double *dq, const double *db, const double *dg,
                                     const double *dd, const double *du, const ssqp_batch_strides *strides,
                                     int32_t *dS, const double *dx0, double *dz, const ssqp_settings *settings,
                                     int64_t *dstatus, int32_t *ddetail, ssqp_stats *dstats, ssqp_trace *dtrace,
                                     int ntrace, double *dlambda, double *dgamma, void *stream) {
    return solve_dev_impl(c, nprob, N, M, J, dV, dA, dG, dq, db, dg, dd, du, strides, dS, dx0, dz, settings, dstatus, ddetail,
                          dstats, dtrace, ntrace, dlambda, dgamma, stream, nullptr);
}

int ssqp_solve_full_batch_dev_f64(ssqp_ctx *c, int nprob, int N, int M, int J, const double *dV, const double *dA,
                                  const double *dG, const double *dq, const double *db, const double *dg,
                                  const double *dd, const double *du, int32_t *dS, double *dz,
                                  const ssqp_settings *settings, const ssqp_settings *settingsLP, int64_t *dstatus,
                                  int32_t *ddetail, ssqp_stats *dstats, double *dlambda, double *dgamma, void *stream) {
    ssqp_settings def;
    ssqp_default_settings(&def);
    const ssqp_settings *lp = settingsLP ? settingsLP : (settings ? settings : &def);   // settingsLP = settings (SSQP.jl:224)
    return solve_dev_impl(c, nprob, N, M, J, dV, dA, dG, dq, db, dg, dd, du, nullptr, dS, nullptr, dz, settings, dstatus, ddetail,
                          dstats, nullptr, 0, dlambda, dgamma, stream, lp);
}

int ssqp_solve_batch_dev_f64(ssqp_ctx *c, int nprob, int N, int M, int J, const double *dV, const double *dA,
                             const double *dG, const double *dq, const double *db, const double *dg,
                             const double *dd, const double *du, int32_t *dS, const double *dx0, double *dz,
                             const ssqp_settings *settings, int64_t *dstatus, int32_t *ddetail,
                             ssqp_stats *dstats, ssqp_trace *dtrace, int ntrace, double *dlambda, double *dgamma,
                             void *stream) {
    return ssqp_solve_batch_strided_dev_f64(c, nprob, N, M, J, dV, dA, dG, dq, db, dg, dd, du, nullptr, dS, dx0, dz,
                                            settings, dstatus, ddetail, dstats, dtrace, ntrace, dlambda, dgamma, stream);
}

int ssqp_generate_V_dev(ssqp_ctx *c, const ssqp_gen_cfg *cfg, uint64_t seed0, int nprob, double *dV, void *stream) {
    if (!c || !cfg || !dV || nprob < 0 || cfg->N <= 0 || cfg->T <= 0) return SSQP_ERR_ARG;
    if (nprob == 0) return SSQP_OK;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return SSQP_ERR_HIP;
    return hip_ok(c, ssqp::launch_genV(nprob, cfg->N, cfg->T, cfg->delta, seed0, dV, (hipStream_t)stream), "genV launch")
               ? SSQP_OK : SSQP_ERR_HIP;
}

// child context number i of c (created on first use, options follow the parent at every call)
static ssqp_ctx *lane_of(ssqp_ctx *c, int i) {
    while ((int)c->lanes.size() <= i) {
        ssqp_ctx *l = nullptr;
        if (ssqp_ctx_create(c->device, &l) != SSQP_OK) return nullptr;
        c->lanes.push_back(l);
    }
    ssqp_ctx *l = c->lanes[(size_t)i];
    l->optWgPerCU = c->optWgPerCU;
    l->optDenseGamma = c->optDenseGamma;
    l->optIncremental = c->optIncremental;
    l->optWaveKernel = c->optWaveKernel;
    l->optWaveQPC = c->optWaveQPC;
    l->optLazyHandover = 1;  // (the batch entry finishes every lane before it copies the results back)
    return l;
}

int ssqp_solve_batch_f64(ssqp_ctx *c, int nprob, int N, int M, int J, const double *V, const double *A,
                         const double *G, const double *q, const double *b, const double *g, const double *d,
                         const double *u, int32_t *S, const double *x0, double *z, const ssqp_settings *settings,
                         int64_t *status, int32_t *detail, ssqp_stats *stats, double *lambda, double *gamma) {
    int rc = check_dims(c, nprob, N, M, J);
    if (rc != SSQP_OK) return rc;
    if (!V || !q || !d || !u || !S || !x0 || !z || !status || (M > 0 && (!A || !b)) || (J > 0 && (!G || !g))) {
        c->err = "null pointer";
        return SSQP_ERR_ARG;
    }
    if (nprob == 0) return SSQP_OK;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return SSQP_ERR_HIP;
    const size_t P = nprob, n = N, m = M, j = J;
    struct Up { DevBuf *buf; const void *src; size_t per; };  // bytes per problem
    const Up ups[] = {{&c->hV, V, n * n * 8}, {&c->hA, A, m * n * 8}, {&c->hG, G, j * n * 8}, {&c->hq, q, n * 8},
                      {&c->hb, b, m * 8},     {&c->hg, g, j * 8},     {&c->hd, d, n * 8},     {&c->hu, u, n * 8},
                      {&c->hS, S, (n + j) * 4}, {&c->hx0, x0, n * 8}};
    for (const Up &up : ups)
        if (!ensure(c, *up.buf, up.per * P)) return SSQP_ERR_ALLOC;
    if (!ensure(c, c->hz, P * n * 8) || !ensure(c, c->hstatus, P * 8) || !ensure(c, c->hdetail, P * 4) ||
        !ensure(c, c->hstats, P * sizeof(ssqp_stats)))
        return SSQP_ERR_ALLOC;
    if ((lambda && !ensure(c, c->hlam, P * (m + j) * 8)) || (gamma && !ensure(c, c->hgam, P * n * 8))) return SSQP_ERR_ALLOC;
    // (multipliers are written for status > 0 only: the device entry every chunk goes through zeroes its part first)
    // The upload of V (N*N*8 bytes per QP over PCIe) dwarfs the solve: the batch goes up in chunks, and every chunk is
    // solved on one of four launch lanes (child contexts) as soon as it has landed -- the solves run behind the
    // upload of the following chunks and beside each other, so the call takes the transfer plus one chunk's solve.
    const size_t vbytes = P * n * n * 8;
    if (c->optPinHost && (c->pinnedPtr != (const void *)V || c->pinnedBytes != vbytes)) {
        // page-lock V where it lies: the upload then runs at the full PCIe rate and truly asynchronously.  The
        // registration (tens of ms for 2 GiB) is kept for the next call with the same array -- a host that solves
        // out of the same buffers again and again pays it once.
        if (c->pinnedPtr) (void)hipHostUnregister(const_cast<void *>(c->pinnedPtr));
        c->pinnedPtr = nullptr;
        if (hipHostRegister(const_cast<double *>(V), vbytes, hipHostRegisterDefault) == hipSuccess) {
            c->pinnedPtr = V;
            c->pinnedBytes = vbytes;
        } else {
            (void)hipGetLastError();  // (not fatal: the pageable path still works)
        }
    }
    int nchunk = 1;
    if (vbytes > ((size_t)64 << 20)) nchunk = (int)((vbytes + ((size_t)256 << 20) - 1) / ((size_t)256 << 20));
    if (nchunk > nprob) nchunk = nprob;
    const int per = (nprob + nchunk - 1) / nchunk;
    const int nlane = nchunk > 1 ? 4 : 1;
    for (int i = 0; i < 4; ++i)
        if (!c->evCopy[i] && !hip_ok(c, hipEventCreateWithFlags(&c->evCopy[i], hipEventDisableTiming), "hipEventCreate"))
            return SSQP_ERR_HIP;
    for (int ch = 0; ch < nchunk; ++ch) {
        const size_t lo = (size_t)ch * per;
        const size_t cnt = (lo + per <= P) ? (size_t)per : P - lo;
        if (cnt == 0) break;
        ssqp_ctx *l = nlane > 1 ? lane_of(c, ch % nlane) : c;
        if (!l) return SSQP_ERR_ALLOC;
        for (const Up &up : ups)
            if (up.per && up.src &&
                !hip_ok(c, hipMemcpyAsync((char *)up.buf->p + lo * up.per, (const char *)up.src + lo * up.per, cnt * up.per,
                                          hipMemcpyHostToDevice, c->stream), "H2D"))
                return SSQP_ERR_HIP;
        hipStream_t ls = l->stream;
        if (l != c) {
            if (!hip_ok(c, hipEventRecord(c->evCopy[ch % 4], c->stream), "hipEventRecord") ||
                !hip_ok(c, hipStreamWaitEvent(ls, c->evCopy[ch % 4], 0), "hipStreamWaitEvent"))
                return SSQP_ERR_HIP;
        }
        auto at = [&](DevBuf &bf, size_t perb) { return (void *)((char *)bf.p + lo * perb); };
        rc = ssqp_solve_batch_dev_f64(l, (int)cnt, N, M, J, (const double *)at(c->hV, n * n * 8),
                                      (const double *)at(c->hA, m * n * 8), (const double *)at(c->hG, j * n * 8),
                                      (const double *)at(c->hq, n * 8), (const double *)at(c->hb, m * 8),
                                      (const double *)at(c->hg, j * 8), (const double *)at(c->hd, n * 8),
                                      (const double *)at(c->hu, n * 8), (int32_t *)at(c->hS, (n + j) * 4),
                                      (const double *)at(c->hx0, n * 8), (double *)at(c->hz, n * 8), settings,
                                      (int64_t *)at(c->hstatus, 8), (int32_t *)at(c->hdetail, 4),
                                      (ssqp_stats *)at(c->hstats, sizeof(ssqp_stats)), nullptr, 0,
                                      lambda ? (double *)at(c->hlam, (m + j) * 8) : nullptr,
                                      gamma ? (double *)at(c->hgam, n * 8) : nullptr, ls);
        if (rc != SSQP_OK) {
            if (l != c) c->err = l->err;
            return rc;
        }
        if (l == c) {  // "lazy_handover" on the caller's context: the owed launch goes out before the results are read
            rc = finish_pending(c);
            if (rc != SSQP_OK) return rc;
        }
    }
    for (int i = 0; i < nlane && nlane > 1; ++i)
        if (i < (int)c->lanes.size()) {
            ssqp_ctx *l = c->lanes[(size_t)i];
            const int rl = ssqp_sync(l, l->stream);
            if (rl != SSQP_OK) {
                c->err = l->err;
                return rl;
            }
        }
    if (!hip_ok(c, hipMemcpyAsync(z, c->hz.p, P * n * 8, hipMemcpyDeviceToHost, c->stream), "D2H") ||
        !hip_ok(c, hipMemcpyAsync(S, c->hS.p, P * (n + j) * 4, hipMemcpyDeviceToHost, c->stream), "D2H") ||
        !hip_ok(c, hipMemcpyAsync(status, c->hstatus.p, P * 8, hipMemcpyDeviceToHost, c->stream), "D2H"))
        return SSQP_ERR_HIP;
    if (detail && !hip_ok(c, hipMemcpyAsync(detail, c->hdetail.p, P * 4, hipMemcpyDeviceToHost, c->stream), "D2H"))
        return SSQP_ERR_HIP;
    if (stats && !hip_ok(c, hipMemcpyAsync(stats, c->hstats.p, P * sizeof(ssqp_stats), hipMemcpyDeviceToHost,
                                           c->stream), "D2H"))
        return SSQP_ERR_HIP;
    if (lambda && (m + j) > 0 && !hip_ok(c, hipMemcpyAsync(lambda, c->hlam.p, P * (m + j) * 8, hipMemcpyDeviceToHost, c->stream), "D2H"))
        return SSQP_ERR_HIP;
    if (gamma && !hip_ok(c, hipMemcpyAsync(gamma, c->hgam.p, P * n * 8, hipMemcpyDeviceToHost, c->stream), "D2H"))
        return SSQP_ERR_HIP;
    if (!hip_ok(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize")) return SSQP_ERR_HIP;
    return SSQP_OK;
}

// ---- a batch kept in HBM: upload once, solve many times (warm starts, sweeps over q / b) ----
int ssqp_problem_upload(ssqp_ctx *c, int nprob, int N, int M, int J, const double *V, const double *A, const double *G,
                        const double *q, const double *b, const double *g, const double *d, const double *u,
                        ssqp_problem **out) {
    if (!out) return SSQP_ERR_ARG;
    *out = nullptr;
    int rc = check_dims(c, nprob, N, M, J);
    if (rc != SSQP_OK) return rc;
    if (nprob <= 0 || !V || !q || !d || !u || (M > 0 && (!A || !b)) || (J > 0 && (!G || !g))) {
        c->err = "null pointer";
        return SSQP_ERR_ARG;
    }
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return SSQP_ERR_HIP;
    ssqp_problem *p = new (std::nothrow) ssqp_problem();
    if (!p) return SSQP_ERR_ALLOC;
    p->ctx = c; p->nprob = nprob; p->N = N; p->M = M; p->J = J;
    const size_t P = nprob, n = N, m = M, j = J;
    struct Up { DevBuf *buf; const void *src; size_t bytes; };
    const Up ups[] = {{&p->V, V, P * n * n * 8}, {&p->A, A, P * m * n * 8}, {&p->G, G, P * j * n * 8}, {&p->q, q, P * n * 8},
                      {&p->b, b, P * m * 8},     {&p->g, g, P * j * 8},     {&p->d, d, P * n * 8},     {&p->u, u, P * n * 8}};
    bool ok = true;
    for (const Up &up : ups) {
        ok = ok && ensure(c, *up.buf, up.bytes);
        if (ok && up.bytes && up.src)
            ok = hip_ok(c, hipMemcpyAsync(up.buf->p, up.src, up.bytes, hipMemcpyHostToDevice, c->stream), "H2D");
    }
    ok = ok && ensure(c, p->S, P * (n + j) * 4) && ensure(c, p->x0, P * n * 8) && ensure(c, p->z, P * n * 8) &&
         ensure(c, p->status, P * 8) && ensure(c, p->detail, P * 4) && ensure(c, p->stats, P * sizeof(ssqp_stats));
    ok = ok && hip_ok(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize");
    if (!ok) {
        (void)ssqp_problem_free(p);
        return SSQP_ERR_ALLOC;
    }
    *out = p;
    return SSQP_OK;
}

int ssqp_problem_free(ssqp_problem *p) {
    if (!p) return SSQP_OK;
    if (p->ctx) (void)hipSetDevice(p->ctx->device);
    for (DevBuf *b : {&p->V, &p->A, &p->G, &p->q, &p->b, &p->g, &p->d, &p->u, &p->S, &p->x0, &p->z, &p->status, &p->detail,
                      &p->stats})
        release(*b);
    delete p;
    return SSQP_OK;
}

int ssqp_problem_set_vector(ssqp_problem *p, int which, const double *data) {
    if (!p || !data || which < 0 || which > 4) return SSQP_ERR_ARG;
    ssqp_ctx *c = p->ctx;
    DevBuf *bufs[] = {&p->q, &p->b, &p->g, &p->d, &p->u};
    const size_t len[] = {(size_t)p->N, (size_t)p->M, (size_t)p->J, (size_t)p->N, (size_t)p->N};
    const size_t bytes = (size_t)p->nprob * len[which] * 8;
    if (bytes == 0) return SSQP_OK;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return SSQP_ERR_HIP;
    if (!hip_ok(c, hipMemcpyAsync(bufs[which]->p, data, bytes, hipMemcpyHostToDevice, c->stream), "H2D") ||
        !hip_ok(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize"))
        return SSQP_ERR_HIP;
    return SSQP_OK;
}

int ssqp_problem_solve(ssqp_problem *p, int32_t *S, const double *x0, double *z, const ssqp_settings *settings,
                       int64_t *status, int32_t *detail, ssqp_stats *stats) {
    if (!p || !S || !x0 || !z || !status) return SSQP_ERR_ARG;
    ssqp_ctx *c = p->ctx;
    const size_t P = p->nprob, n = p->N, j = p->J;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return SSQP_ERR_HIP;
    if (!hip_ok(c, hipMemcpyAsync(p->S.p, S, P * (n + j) * 4, hipMemcpyHostToDevice, c->stream), "H2D") ||
        !hip_ok(c, hipMemcpyAsync(p->x0.p, x0, P * n * 8, hipMemcpyHostToDevice, c->stream), "H2D"))
        return SSQP_ERR_HIP;
    int rc = ssqp_solve_batch_dev_f64(c, p->nprob, p->N, p->M, p->J, (const double *)p->V.p, (const double *)p->A.p,
                                      (const double *)p->G.p, (const double *)p->q.p, (const double *)p->b.p,
                                      (const double *)p->g.p, (const double *)p->d.p, (const double *)p->u.p,
                                      (int32_t *)p->S.p, (const double *)p->x0.p, (double *)p->z.p, settings,
                                      (int64_t *)p->status.p, (int32_t *)p->detail.p, (ssqp_stats *)p->stats.p, nullptr, 0,
                                      nullptr, nullptr, c->stream);
    if (rc == SSQP_OK) rc = finish_pending(c);  // ("lazy_handover": the owed launch goes out before the results are read)
    if (rc != SSQP_OK) return rc;
    if (!hip_ok(c, hipMemcpyAsync(z, p->z.p, P * n * 8, hipMemcpyDeviceToHost, c->stream), "D2H") ||
        !hip_ok(c, hipMemcpyAsync(S, p->S.p, P * (n + j) * 4, hipMemcpyDeviceToHost, c->stream), "D2H") ||
        !hip_ok(c, hipMemcpyAsync(status, p->status.p, P * 8, hipMemcpyDeviceToHost, c->stream), "D2H"))
        return SSQP_ERR_HIP;
    if (detail && !hip_ok(c, hipMemcpyAsync(detail, p->detail.p, P * 4, hipMemcpyDeviceToHost, c->stream), "D2H"))
        return SSQP_ERR_HIP;
    if (stats && !hip_ok(c, hipMemcpyAsync(stats, p->stats.p, P * sizeof(ssqp_stats), hipMemcpyDeviceToHost, c->stream), "D2H"))
        return SSQP_ERR_HIP;
    return hip_ok(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize") ? SSQP_OK : SSQP_ERR_HIP;
}

int ssqp_phase1_batch_dev_f64(ssqp_ctx *c, int nprob, int N, int M, int J, const double *dA, const double *dG,
                              const double *db, const double *dg, const double *dd, const double *du,
                              const ssqp_settings *settingsLP, double *dx0, int32_t *dS, int32_t *dstatus, void *stream) {
    if (!c) return SSQP_ERR_ARG;
    if (nprob < 0 || N <= 0 || M < 0 || J < 0 || !dd || !du || !dx0 || !dS || !dstatus || (M > 0 && (!dA || !db)) ||
        (J > 0 && (!dG || !dg))) {
        c->err = "bad argument";
        return SSQP_ERR_ARG;
    }
    if (nprob == 0) return SSQP_OK;
    ssqp_settings def;
    ssqp_default_settings(&def);
    const ssqp_settings *st = settingsLP ? settingsLP : &def;
    if (st->rule != 0) {
        c->err = "only rule = :Dantzig is implemented for Phase-1";
        return SSQP_ERR_UNSUPPORTED;
    }
    if (M + J > ssqp::NT || ssqp::phase1_lds_bytes(M, J) > (size_t)ssqp::LDS_BYTES) {  // (one thread per basic row)
        c->err = "M + J too large for the GPU Phase-1 (the basis inverse does not fit in LDS): use ssqp_phase1_batch_f64";
        return SSQP_ERR_UNSUPPORTED;
    }
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return SSQP_ERR_HIP;
    const size_t wd = ssqp::phase1_ws_doubles(N, M, J), wi = ssqp::phase1_ws_ints(N, M, J);
    hipStream_t s = (hipStream_t)stream;
    // the one-wavefront-per-QP kernel takes the shapes it is built for (M + J <= 11, N + J + M + J <= 576); what it leaves
    // on its list -- QPs with free variables -- goes to the workgroup kernel, which a bounded grid works off (an empty list:
    // that grid exits at once)
    const bool wave = c->optPhase1Wave && ssqp::phase1_wave_applies(N, M, J);
    const unsigned int *listCount = nullptr;
    const int *list = nullptr;
    if (wave) {
        if (!ensure(c, c->p1queue, 64) || !ensure(c, c->p1list, (size_t)nprob * 4)) return SSQP_ERR_ALLOC;
        if (!hip_ok(c, hipMemsetAsync(c->p1queue.p, 0, 64, s), "hipMemsetAsync")) return SSQP_ERR_HIP;
        if (!hip_ok(c, ssqp::launch_phase1_wave(nprob, N, M, J, dA, dG, db, dg, dd, du, st->tol, dx0, dS, dstatus,
                                                (unsigned int *)c->p1queue.p, (int *)c->p1list.p, s), "phase-1 wave launch"))
            return SSQP_ERR_HIP;
        listCount = (const unsigned int *)c->p1queue.p;
        list = (const int *)c->p1list.p;
    }
    // (the workspaces are indexed by problem id: a listed QP may be any of them)
    if (!ensure(c, c->p1ws, (size_t)nprob * wd * 8) || !ensure(c, c->p1wsInt, (size_t)nprob * wi * 4)) return SSQP_ERR_ALLOC;
    return hip_ok(c, ssqp::launch_phase1(nprob, N, M, J, dA, dG, db, dg, dd, du, st->tol, dx0, dS, dstatus,
                                         (double *)c->p1ws.p, wd, (int *)c->p1wsInt.p, wi, listCount, list, list ? 64 : 0, nullptr, s),
                  "phase-1 launch") ? SSQP_OK : SSQP_ERR_HIP;
}

int ssqp_solve_batch_multi_f64(ssqp_ctx *const *ctxs, int nctx, int nprob, int N, int M, int J, const double *V,
                               const double *A, const double *G, const double *q, const double *b, const double *g,
                               const double *d, const double *u, int32_t *S, const double *x0, double *z,
                               const ssqp_settings *settings, int64_t *status, int32_t *detail, ssqp_stats *stats) {
    if (!ctxs || nctx <= 0 || nprob < 0) return SSQP_ERR_ARG;
    for (int k = 0; k < nctx; ++k)
        if (!ctxs[k]) return SSQP_ERR_ARG;
    if (nprob == 0) return SSQP_OK;
    // contiguous blocks: context r owns problems [r*ceil(P/G), (r+1)*ceil(P/G))  (SURVEY.md section 8e); the QPs are
    // independent, so there is no exchange between the shards -- every shard's results land in the caller's arrays
    const int per = (nprob + nctx - 1) / nctx;
    std::vector<int> rcs((size_t)nctx, SSQP_OK);
    std::vector<std::thread> th;
    const size_t n = (size_t)N, m = (size_t)M, j = (size_t)J;
    for (int r = 0; r < nctx; ++r) {
        const int lo = r * per < nprob ? r * per : nprob;
        const int hi = lo + per < nprob ? lo + per : nprob;
        if (hi <= lo) continue;
        th.emplace_back([=, &rcs]() {
            const size_t o = (size_t)lo;
            rcs[(size_t)r] = ssqp_solve_batch_f64(ctxs[r], hi - lo, N, M, J, V + o * n * n, A ? A + o * m * n : nullptr,
                                                  G ? G + o * j * n : nullptr, q + o * n, b ? b + o * m : nullptr,
                                                  g ? g + o * j : nullptr, d + o * n, u + o * n, S + o * (n + j),
                                                  x0 + o * n, z + o * n, settings, status + o,
                                                  detail ? detail + o : nullptr, stats ? stats + o : nullptr, nullptr, nullptr);
        });
    }
    for (std::thread &t : th) t.join();
    for (int r = 0; r < nctx; ++r)
        if (rcs[(size_t)r] != SSQP_OK) return rcs[(size_t)r];
    return SSQP_OK;
}

int ssqp_solve_f64(ssqp_ctx *c, int N, int M, int J, const double *V, const double *A, const double *G,
                   const double *q, const double *b, const double *g, const double *d, const double *u, int32_t *S,
                   const double *x0, double *z, const ssqp_settings *settings, int64_t *status, int32_t *detail) {
    return ssqp_solve_batch_f64(c, 1, N, M, J, V, A, G, q, b, g, d, u, S, x0, z, settings, status, detail, nullptr, nullptr, nullptr);
}

int ssqp_solve_full_f64(ssqp_ctx *c, int N, int M, int J, const double *V, const double *A, const double *G,
                        const double *q, const double *b, const double *g, const double *d, const double *u, int mc,
                        int32_t *S, double *z, const ssqp_settings *settings, const ssqp_settings *settingsLP,
                        int64_t *status, int32_t *detail) {
    if (!c || !S || !z || !status || N <= 0) return SSQP_ERR_ARG;
    if (detail) *detail = SSQP_DETAIL_NONE;
    if (mc <= 0) {  // SSQP.jl:226-228
        for (int k = 0; k < N; ++k) {
            z[k] = 0.0;
            S[k] = SSQP_DN;
        }
        *status = -1;
        if (detail) *detail = SSQP_DETAIL_MODEL;
        return SSQP_OK;
    }
    std::vector<double> x0(N);
    int32_t st1 = 0;
    int rc = ssqp_phase1_f64(N, M, J, A, G, b, g, d, u, settingsLP ? settingsLP : settings, x0.data(), S, &st1);
    if (rc != SSQP_OK) return rc;
    if (st1 <= 0) {  // SSQP.jl:230-232
        std::memcpy(z, x0.data(), sizeof(double) * (size_t)N);
        *status = st1;
        if (st1 < 0 && detail) *detail = SSQP_DETAIL_SINGULAR_LU;
        return SSQP_OK;
    }
    return ssqp_solve_f64(c, N, M, J, V, A, G, q, b, g, d, u, S, x0.data(), z, settings, status, detail);
}

}  // extern "C"
