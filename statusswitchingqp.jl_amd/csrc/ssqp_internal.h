// ssqp_internal.h -- shared between the kernels (ssqp_kernels.hip) and the
// C-ABI host layer (ssqp_api.hip).  Not part of the public interface.
#ifndef SSQP_INTERNAL_H
#define SSQP_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "ssqp_hip.h"

namespace ssqp {

constexpr int NT = 256;          // threads per workgroup (4 wavefronts of 64: one per SIMD)
constexpr int NW = NT / 64;      // wavefronts per workgroup
constexpr int MAXPT = 8;         // per-thread slots over the free list
constexpr int MAXN = NT * MAXPT; // largest N the in-kernel loop accepts (2048)
// at most 2 workgroups per CU (2 waves per SIMD -> 256 VGPRs each): one streams V while the other is in its
// latency-bound phases (factor updates, reductions)
constexpr int MAX_WG_PER_CU = 2;
constexpr int LDS_BYTES = 160 * 1024;  // gfx950: 160 KiB per CU, one workgroup may use all

struct SolveParams {
    int nprob, N, M, J, MJ;
    const double *V;    // nprob x (N x N) column-major, symmetric
    const double *Ct;   // nprob x (MJ x N): row r of [A;G] contiguous
    const double *rhs;  // nprob x MJ: [b; g]
    const double *q, *d, *u, *x0;
    size_t sV, sCt, sRhs, sq, sd, su;  // per-problem element strides (0 = shared by the batch)
    int32_t *S;
    double *z;
    int64_t *status;
    int32_t *detail;
    ssqp_stats *stats;
    ssqp_trace *trace;
    int ntrace;
    // multipliers of the last pass (SSQP.jl:351-352, 149-171), by row / variable id; either may be null
    double *lamOut;   // nprob x MJ
    double *gamOut;   // nprob x N
    int maxIter;
    double tol, tolG;
    unsigned int *queue;     // work counter, zeroed before the launch
    double *gscratch;        // per-workgroup global arena (used when the LDS arena is too small)
    size_t gscratchStride;   // doubles per workgroup
    int arenaCap;            // doubles in the LDS arena
    int denseGamma;          // 1: the gamma pass reads every column of V (dense formulation, for roofline runs)
    int incremental;         // 1: keep the LDL' factor across passes (append/delete) instead of refactoring
    // ---- hand-over between the two solve kernels (ssqp_wave.hip -> ssqp_kernels.hip) ----
    // The wavefront kernel leaves a QP it cannot finish (free set or shape outside its limits) with its current
    // (z, S) written to P.z / P.S, the passes done so far in fbIter[prob] and the problem id appended to fbList;
    // the workgroup kernel then continues those QPs from exactly that state: the loop has no other state
    // (SSQP.jl:237-377 recomputes everything from (z, S) in every pass).
    unsigned int *fbCount;   // number of entries of fbList (device counter, zeroed before the launch)
    int *fbList;             // problem ids handed over
    long long *fbIter;       // per problem: loop passes already done
    int resume;              // 1 = take the QPs of a hand-over list (start from P.z, fbIter) instead of 0..nprob-1: the
                             // workgroup kernel reads (fbCount, fbList); the big-factor wavefront kernel reads
                             // (resumeCount, resumeList) and hands what it cannot finish over through (fbCount, fbList)
    const unsigned int *resumeCount;
    const int *resumeList;
    double *wscratch;        // wavefront kernel: per-wavefront global scratch
    size_t wscratchStride;   // doubles per wavefront
    int waveLdsBytes;        // wavefront kernel: dynamic LDS per wavefront
    int waveRC;              // wavefront kernel: row capacity of the kept factor
};

// Offsets of the LDS carve-up.  Double-typed regions first (offsets in
// doubles), then the integer regions (offsets in bytes).
struct LdsLayout {
    int z, zm, gam, hq, arena, bE, aL, tv, dcol, lin, bEall, red;  // in doubles
    int S_bytes, ired_bytes, pos_bytes, idx_bytes, perm_bytes, rowsE_bytes, ra_bytes, iO_bytes, fpos_bytes, ordl_bytes, ytag_bytes, evt_bytes;
    int total_bytes;
};

__host__ __device__ inline int align_up(int x, int a) { return (x + a - 1) / a * a; }

// bytes of everything except the arena
__host__ __device__ inline int lds_fixed_bytes(int N, int M, int J) {
    const int MJ1 = align_up(M + J + 1, 2);
    int dbl = 4 * align_up(N, 2) + 6 * MJ1 + 2 * NW;  // z, zm, gam, hq | bE, aL, tv, dcol, lin, bEall | red
    int bytes = dbl * 8;
    bytes += align_up(4 * (N + J), 8);           // S
    bytes += align_up(4 * (2 * NW + 16 + NW * MAXPT), 8);  // ired (+ per-chunk wave counts of the compaction)
    bytes += 4 * align_up(2 * (N + 2), 8);       // pos, idx, fpos, ordl
    bytes += align_up(2 * (N + M + J + 4), 8);   // perm (column list of the AXPY pass: N columns + constraint rows + q)
    bytes += 3 * align_up(2 * (M + J + 2), 8);   // rowsE, ra, iO
    bytes += 32;                                 // ytag (row ids of the kept border columns)
    bytes += 16 * 8 + 16 * 4;                    // evt: variables that entered B with a nonzero value in this pass
    return bytes;
}

__host__ __device__ inline LdsLayout lds_layout(int N, int M, int J, int arenaCap) {
    LdsLayout l;
    const int Np = align_up(N, 2), MJ1 = align_up(M + J + 1, 2);
    int o = 0;
    l.z = o; o += Np;
    l.zm = o; o += Np;
    l.gam = o; o += Np;
    l.hq = o; o += Np;
    l.bE = o; o += MJ1;
    l.aL = o; o += MJ1;
    l.tv = o; o += MJ1;
    l.dcol = o; o += MJ1;
    l.lin = o; o += MJ1;
    l.bEall = o; o += MJ1;
    l.red = o; o += 2 * NW;
    l.arena = o; o += align_up(arenaCap, 2);
    int b = o * 8;
    l.S_bytes = b; b += align_up(4 * (N + J), 8);
    l.ired_bytes = b; b += align_up(4 * (2 * NW + 16 + NW * MAXPT), 8);
    l.pos_bytes = b; b += align_up(2 * (N + 2), 8);
    l.idx_bytes = b; b += align_up(2 * (N + 2), 8);
    l.perm_bytes = b; b += align_up(2 * (N + M + J + 4), 8);
    l.rowsE_bytes = b; b += align_up(2 * (M + J + 2), 8);
    l.ra_bytes = b; b += align_up(2 * (M + J + 2), 8);
    l.iO_bytes = b; b += align_up(2 * (M + J + 2), 8);
    l.fpos_bytes = b; b += align_up(2 * (N + 2), 8);
    l.ordl_bytes = b; b += align_up(2 * (N + 2), 8);
    l.ytag_bytes = b; b += 32;
    l.evt_bytes = b; b += 16 * 8 + 16 * 4;
    l.total_bytes = b;
    return l;
}

// doubles one workgroup may need in the global arena (K = N, W0 = M+J)
inline size_t global_arena_doubles(int N, int M, int J) {
    const size_t K = N, W0 = M + J, R = K + W0 + 1;
    const size_t packed = R * (R + 1) / 2;
    const size_t x = W0 * (K + 1) + W0;
    const size_t ls = K * W0 + W0 * W0 + W0;
    size_t f = (packed + R > ls ? packed + R : ls) + 64;
    if (f < (size_t)NW * N + 64) f = (size_t)NW * N + 64;   // staging of the AXPY partial vectors
    // kept-factor engine in the global arena: staging in front, then Y (12 columns), 1/d, factor for up to 256 rows
    const size_t rcg = N < 256 ? N : 256;
    const size_t eng = (size_t)NW * N + 64 + 13 * rcg + rcg * (rcg + 1) / 2 + 64;
    if (f < eng) f = eng;
    return (x > f ? x : f) + 64;
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the kernel (per device), not to a launch: two host threads that
// set different sizes before their launches would race.  It is set ONCE per kernel and device, to the whole LDS; the
// size a launch asks for is what decides the residency.  `doneMask`: one static word per kernel, bit = device.
inline hipError_t allow_full_lds(const void *fn, unsigned long long *doneMask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(doneMask, __ATOMIC_ACQUIRE) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    __atomic_fetch_or(doneMask, bit, __ATOMIC_RELEASE);
    return hipSuccess;
}

void launch_prep(int nct, int nrhs, int N, int M, int J, const double *A, const double *G, const double *b,
                 const double *g, size_t sA, size_t sG, size_t sb, size_t sg, double *Ct, double *rhs,
                 hipStream_t stream);
hipError_t launch_genV(int nprob, int N, int T, double delta, unsigned long long seed0, double *V, hipStream_t stream);
hipError_t launch_solve(const SolveParams &P, int grid, size_t ldsBytes, int wgPerCU, hipStream_t stream);

// ---- wavefront kernel (ssqp_wave.hip): one 64-lane wavefront per QP ----
constexpr int WAVE_MAXN = 512;   // dense N-vectors live in registers: 8 doubles per lane
constexpr int WAVE_MJ = 11;      // constraint rows carried per free variable (M + J <= 11)
bool wave_kernel_applies(int N, int M, int J);
// LDS bytes one wavefront needs for a kept factor of `rc` rows in LDS (rc <= 0: rows >= 64 in global scratch)
int wave_lds_bytes(int rc);
int wave_lds_bytes_big();  // LDS bytes of the big-factor build (kept factor rows < 64, Schur block, the column ring)
// doubles of global scratch per wavefront of build `variant`
size_t wave_scratch_doubles(int variant);
constexpr int WAVE_BIG_ROWS = 252;  // row capacity of the big-factor build (four row slots of 64)
// variant 0: four QPs per CU (one wavefront per SIMD, every row in LDS); 1: eight per CU (rows >= 64 in global scratch);
// 2: the big-factor build (four row slots, up to WAVE_BIG_ROWS free variables, rows >= 64 in global scratch)
hipError_t launch_solve_wave(const SolveParams &P, int grid, int variant, hipStream_t stream);

// ---- Phase-1 on the GPU (ssqp_phase1.hip): one workgroup per QP
size_t phase1_ws_doubles(int N, int M, int J);
size_t phase1_ws_ints(int N, int M, int J);
size_t phase1_lds_bytes(int M, int J);
// single-launch solveQP(Q): where the workgroup Phase-1 kernel sends a QP of its list on into the loop (a hand-over at pass 0)
struct Phase1Handover {
    unsigned int *count;
    int *list;
    long long *iter;
    double *z;
    int64_t *status;
    int32_t *detail;
    ssqp_stats *stats;
};
hipError_t launch_phase1(int nprob, int N, int M, int J, const double *A, const double *G, const double *b, const double *g,
                         const double *d, const double *u, double tol, double *x0, int32_t *S, int32_t *status, double *ws,
                         size_t wsStride, int *wsInt, size_t wsIntStride, const unsigned int *listCount, const int *list, int gridCap,
                         const Phase1Handover *ho, hipStream_t stream);
// ---- Phase-1, one wavefront per QP (ssqp_phase1_wave.hip): M + J <= 11 rows and N + J + M + J <= 576 columns; a QP with a
// free variable is left on (fbCount, fbList) for the workgroup kernel
bool phase1_wave_applies(int N, int M, int J);
hipError_t launch_phase1_wave(int nprob, int N, int M, int J, const double *A, const double *G, const double *b, const double *g,
                              const double *d, const double *u, double tol, double *x0, int32_t *S, int32_t *status,
                              unsigned int *fbCount, int *fbList, hipStream_t stream);

// ---- solveQP(Q) in one launch (ssqp_wave.hip built with -DSSQP_FULL): the wavefront Phase-1 in front of the four-per-CU
// build of the loop; P as for launch_solve_wave(variant 0), x0 = the vertex buffer (P.x0 must point to it too)
hipError_t launch_solve_full(const SolveParams &P, int grid, const double *A, const double *G, const double *b, const double *g,
                             double tolLP, double *x0, int32_t *p1status, unsigned int *p1Count, int *p1List, hipStream_t stream);

}  // namespace ssqp
#endif
