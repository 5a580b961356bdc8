// ssqp_wave.hip -- gfx950 (MI355X / CDNA4): the active-set loop with ONE 64-lane WAVEFRONT per QP.
//
// The headline workload (BASELINE.json: batches of N = 512 portfolio QPs) runs with small free sets (K = |F| is
// ~19 on average, at most ~80) and at most 11 constraint rows: every pass of solveQP(Q,S,x0) (reference:
// src/SSQP.jl:237-377) is a few thousand dependent scalar-ish operations, not a matrix problem.  A workgroup of
// four wavefronts with a barrier between its ~20 phases (ssqp_kernels.hip) spends most of its time waiting; this
// kernel gives every QP one wavefront instead:
//   * no workgroup barrier anywhere -- all exchanges are DPP / v_readlane / ds_bpermute inside the wavefront;
//   * four (or more) independent QPs per CU, one or two per SIMD, so every SIMD always has its own chain to run;
//   * everything that belongs to a free variable lives in the REGISTERS of "its" lane (row r of the kept factor
//     = lane r & 63 of slot r >> 6): index, rank, z, bounds, pivot, the forward-substituted border rows Y;
//   * N-vectors (hq = V[:,nz(zB)] zB + q, S, gamma) live in registers too, 8 doubles per lane (N <= 512);
//   * LDS holds only the kept LDL' factor of V[F,F], the 12 x 12 Schur block and the Gram matrix of the border rows;
//   * the Schur block H = [AE; c'] V_FF^-1 [AE' c] is kept across passes for ALL rows of [A;G]: an appended free
//     variable adds one rank-1 term, a deleted one removes one (H -= g g'/m, the block-inverse identity), so a
//     change of the active inequality set costs nothing and no pass re-forms H from the factor.
// Pass structure and every decision rule (thresholds, tie rules, event order) are those of the reference, cited
// line by line below; the arithmetic that only feeds threshold tests far above rounding noise (dot products,
// factor + solve instead of explicit inverses) is free to use another order, as in ssqp_kernels.hip.
//
// Shapes this kernel takes: N even, N <= 512, M + J <= 11.  A QP whose free set outgrows the factor capacity is
// HANDED OVER: its current (z, S) and pass count are written out and the workgroup kernel continues from exactly
// that state (the loop has no other state: the reference recomputes everything from (z, S) in every pass).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssqp_hip.h"
#include "ssqp_internal.h"
#include "ssqp_device.h"
#ifdef SSQP_FULL
#include "ssqp_phase1_wave.h"   // (the single-launch solveQP(Q): Phase-1 in front of the loop, SSQP_WAVE_VARIANT 0 only)
#endif

// -DSSQP_WAVE_LEAN: the builds for launches that ask for neither statistics nor a trace (both optional pointers of the
// boundary): the byte / flop accounting and the per-pass trace records are not compiled in -- fewer live scalars in a
// kernel that spills hundreds of them.  Same decisions, same results; the API picks the build by the pointers it is given.
#ifdef SSQP_WAVE_LEAN
#define ACCT(stmt) do { } while (0)
#define SSQP_LEAN_BUILD 1
#else
#define ACCT(stmt) do { stmt; } while (0)
#define SSQP_LEAN_BUILD 0
#endif

namespace ssqp {

// ---- diagnostic build only (-DSSQP_PHASE_PROFILE): cycles per phase of the wavefront kernel ----
#ifdef SSQP_PHASE_PROFILE
// (accumulated in registers and flushed once per QP: an atomic per stamp would sit in front of every later load of
// the phase -- vector memory operations complete in order -- and charge its own round trip to that phase)
static __device__ unsigned long long g_wphase[64];  // (one per build of the kernel)
#define WPH_DECL unsigned long long wph_t = __builtin_amdgcn_s_memtime()
#define WPH(slot)                                                        \
    do {                                                                 \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();      \
        C.ph[(slot)] += t_ - wph_t;                                      \
        C.pn[(slot)] += 1;                                               \
        wph_t = __builtin_amdgcn_s_memtime();                            \
    } while (0)
#else
#define WPH_DECL do { } while (0)
#define WPH(slot) do { } while (0)
#endif

// Everything below has internal linkage: the three builds of this file define the same names with different row-slot
// counts (struct layouts differ), and each build's code must stay in its own translation unit.
namespace {
namespace wv {

constexpr int MJX = WAVE_MJ;  // constraint rows carried (M + J <= MJX)
constexpr int NR = MJX + 1;   // border columns: every row of [A;G] and c
constexpr int CC = MJX;       // index of the c column
constexpr int NCH = 4;        // dense layout: element i sits in lane (i >> 1) & 63, chunk i >> 7, half i & 1
constexpr int KSLOT = 64;     // rows per register slot
// row slots a build carries in registers: 2 (up to 127 rows) for the two builds the headline shapes run on, 4 (up to 255
// rows) for the big-factor build (SSQP_WAVE_VARIANT 2) that takes over the QPs whose free set outgrows those
#ifndef SSQP_WAVE_VARIANT
#define SSQP_WAVE_VARIANT 0
#endif
#if SSQP_WAVE_VARIANT == 2
constexpr int NSL = 4;
#else
constexpr int NSL = 2;
#endif
// only the big-factor build is ever launched on a hand-over list (P.resume): the other builds do not carry the code
constexpr bool CAN_RESUME = NSL > 2;
// (rounded up to whole 128-byte lines: the factor storage behind it, and with it every 1 KiB DMA piece and every double2
//  access, then starts on a cache-line boundary in every wavefront's scratch -- the per-wavefront stride is a multiple too)
[[maybe_unused]] constexpr int WAVE_LS_DOUBLES = (128 * MJX + MJX * MJX + MJX + 21 + 15) / 16 * 16;  // global scratch of the purged-row least squares
[[maybe_unused]] constexpr int WAVE_LS_DOUBLES_BIG = (256 * MJX + MJX * MJX + MJX + 21 + 15) / 16 * 16;  // ... of the big-factor build (up to 256 rows)
static_assert(WAVE_LS_DOUBLES % 16 == 0 && WAVE_LS_DOUBLES_BIG % 16 == 0, "line-aligned factor storage");
constexpr double INF = __builtin_huge_val();

// compile-time loop: the body sees its index as a constant, so every register-array index is static whatever
// the optimiser decides about unrolling
template <int I>
struct IC {
    static constexpr int value = I;
};
template <int B, int E, class F>
__device__ __forceinline__ void sfor(F &&f) {
    if constexpr (B < E) {
        f(IC<B>{});
        sfor<B + 1, E>(f);
    }
}
#define SFOR_BODY(name) [&](auto name##_c) __attribute__((always_inline))
#define SFOR_IDX(name) constexpr int name = decltype(name##_c)::value

// ------------------------------------------------------------------ lane helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

template <int SL>
__device__ __forceinline__ double rbcast(const double (&v)[NSL], int r) {  // value of row r (uniform r)
    if (SL == 1) return readlane_f64(v[0], r);
    if (SL == 2) {
        const double a = readlane_f64(v[0], r & 63), b = readlane_f64(v[1], r & 63);
        return (r < KSLOT) ? a : b;
    }
    // more slots: r is uniform, so the slot is picked by a scalar branch and one pair of v_readlane runs
    double x = 0.0;
    const int l = r & 63;
    sfor<0, SL>(SFOR_BODY(t) {
        SFOR_IDX(t);
        if ((r >> 6) == t) x = readlane_f64(v[t], l);
    });
    return x;
}
template <int SL>
__device__ __forceinline__ int rbcast_i(const int (&v)[NSL], int r) {
    if (SL == 1) return __builtin_amdgcn_readlane(v[0], r);
    if (SL == 2) {
        const int a = __builtin_amdgcn_readlane(v[0], r & 63), b = __builtin_amdgcn_readlane(v[1], r & 63);
        return (r < KSLOT) ? a : b;
    }
    int x = 0;
    const int l = r & 63;
    sfor<0, SL>(SFOR_BODY(t) {
        SFOR_IDX(t);
        if ((r >> 6) == t) x = __builtin_amdgcn_readlane(v[t], l);
    });
    return x;
}
template <int SL>
__device__ __forceinline__ void set_row(double (&v)[NSL], int r, double x) {
    const int lane = lane_id();
#pragma unroll
    for (int t = 0; t < SL; ++t) v[t] = (lane + KSLOT * t == r) ? x : v[t];
}
template <int SL>
__device__ __forceinline__ void set_row_i(int (&v)[NSL], int r, int x) {
    const int lane = lane_id();
#pragma unroll
    for (int t = 0; t < SL; ++t) v[t] = (lane + KSLOT * t == r) ? x : v[t];
}
// value held by lane `src` (per-lane src): LDS crossbar, no memory
__device__ __forceinline__ int bperm_i(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
__device__ __forceinline__ double bperm_f64(double v, int src) {
    const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
// NV wavefront sums at once (NV a multiple of 4): a butterfly that HALVES the number of vectors a lane carries in each
// of its first two steps (lane pairs, then quads, exchange what the partner keeps), so the 64-lane reduction of NV
// vectors costs about NV DPP-adds instead of 6 NV, and the dependent chain is one reduction deep instead of NV.
// Afterwards vector w = i + (NV/4) b1 + (NV/2) b0 sits in slot i of every lane of class (b0, b1) = (lane & 1,
// (lane >> 1) & 1); out[w] is handed to all lanes by v_readlane.
template <int NV>
__device__ __forceinline__ void wave_sum_multi(const double (&v)[NV], double (&out)[NV]) {
    static_assert(NV % 4 == 0, "NV must be a multiple of 4");
    constexpr int H = NV / 2, Q = NV / 4;
    const int lane = lane_id();
    const bool b0 = lane & 1, b1 = lane & 2;
    double r[H], q[Q];
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const double keep = b0 ? v[i + H] : v[i], send = b0 ? v[i] : v[i + H];
        r[i] = keep + dpp_f64<DPP_XOR1>(send);
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const double keep = b1 ? r[i + Q] : r[i], send = b1 ? r[i] : r[i + Q];
        q[i] = keep + dpp_f64<DPP_XOR2>(send);
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        q[i] += dpp_f64<0x124>(q[i]);  // row_ror:4  (lanes of one class inside a 16-lane row)
        q[i] += dpp_f64<0x128>(q[i]);  // row_ror:8
        q[i] += bperm_f64(q[i], lane ^ 16);
        q[i] += bperm_f64(q[i], lane ^ 32);
    }
#pragma unroll
    for (int w = 0; w < NV; ++w) {
        const int wb0 = w / H, rem = w % H, wb1 = rem / Q, i = rem % Q;
        out[w] = readlane_f64(q[i], wb0 + 2 * wb1);
    }
}

// lane i takes the value of lane i + 1 (DPP wave_shl:1; lane 63 keeps its own value)
__device__ __forceinline__ double lane_next_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int lane_next_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xF, 0xF, false); }
// lane i takes the value of lane i - 1 (DPP wave_shr:1; lane 0 gets `fill`)
__device__ __forceinline__ double lane_prev_f64(double v, double fill) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x138, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x138, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// one step of an inclusive scan over the 64 lanes: the value of the lane the step combines with, `fill` where there is
// none.  Steps 0..3: row_shr:1,2,4,8 inside the 16-lane rows; step 4: row_bcast:15 (the last lane of rows 0 and 2 to
// every lane of rows 1 and 3); step 5: row_bcast:31 (lane 31 to rows 2 and 3).
template <int STEP>
__device__ __forceinline__ double scan_src_f64(double v, double fill) {
    constexpr int ctrl = (STEP == 0) ? 0x111 : (STEP == 1) ? 0x112 : (STEP == 2) ? 0x114 : (STEP == 3) ? 0x118
                       : (STEP == 4) ? 0x142 : 0x143;
    constexpr int rowmask = (STEP < 4) ? 0xF : (STEP == 4) ? 0xA : 0xC;
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, ctrl, rowmask, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, ctrl, rowmask, 0xF, false);
    return __hiloint2double(hi, lo);
}

// rows p.. move up by one (row r takes row r+1) in a per-row register set
template <int SL>
__device__ __forceinline__ void shift_up(double (&v)[NSL], int p) {
    const int lane = lane_id();
    double n[NSL], first[NSL];
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        n[t] = lane_next_f64(v[t]);
        first[t] = (t > 0) ? readlane_f64(v[t], 0) : 0.0;
    }
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const double m = (t + 1 < SL && lane == 63) ? first[t + 1] : n[t];  // (lane 63 takes the first row of the next slot)
        v[t] = (lane + KSLOT * t >= p) ? m : v[t];
    }
}
template <int SL>
__device__ __forceinline__ void shift_up_i(int (&v)[NSL], int p) {
    const int lane = lane_id();
    int n[NSL], first[NSL];
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        n[t] = lane_next_i32(v[t]);
        first[t] = (t > 0) ? __builtin_amdgcn_readlane(v[t], 0) : 0;
    }
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int m = (t + 1 < SL && lane == 63) ? first[t + 1] : n[t];
        v[t] = (lane + KSLOT * t >= p) ? m : v[t];
    }
}

// ------------------------------------------------------------------ dense N-vectors in registers
__device__ __forceinline__ int dk_of(int i) { return ((i >> 7) << 1) | (i & 1); }
__device__ __forceinline__ int dl_of(int i) { return (i >> 1) & 63; }
// element i (uniform) of a dense vector.  The lane's eight components are broadcast first and the choice is made
// among the broadcast values: a select chain over the components themselves is folded by the optimiser into ONE
// load at a computed offset, and a computed offset into the state keeps the whole state out of registers.
__device__ __forceinline__ double dense_get(const double2 (&v)[NCH], int i) {
    const int k = dk_of(i), l = dl_of(i);
    double x = readlane_f64(v[0].x, l);
    const double c1 = readlane_f64(v[0].y, l), c2 = readlane_f64(v[1].x, l), c3 = readlane_f64(v[1].y, l);
    const double c4 = readlane_f64(v[2].x, l), c5 = readlane_f64(v[2].y, l), c6 = readlane_f64(v[3].x, l);
    const double c7 = readlane_f64(v[3].y, l);
    x = (k == 1) ? c1 : x;
    x = (k == 2) ? c2 : x;
    x = (k == 3) ? c3 : x;
    x = (k == 4) ? c4 : x;
    x = (k == 5) ? c5 : x;
    x = (k == 6) ? c6 : x;
    x = (k == 7) ? c7 : x;
    return x;
}
__device__ __forceinline__ void dense_set(double2 (&v)[NCH], int i, double x) {  // uniform i
    const int k = dk_of(i);
    const bool me = lane_id() == dl_of(i);
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        v[m].x = (me && k == 2 * m) ? x : v[m].x;
        v[m].y = (me && k == 2 * m + 1) ? x : v[m].y;
    }
}
// per-lane index: v[idx] for every lane (row lanes fetching their variable's entry)
__device__ __forceinline__ double dense_gather(const double2 (&v)[NCH], int idx) {
    const int k = dk_of(idx), l = dl_of(idx);
    double r = 0.0;
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const double a = bperm_f64(v[m].x, l), b = bperm_f64(v[m].y, l);
        r = (k == 2 * m) ? a : r;
        r = (k == 2 * m + 1) ? b : r;
    }
    return r;
}
__device__ __forceinline__ int st_get(unsigned Sp, int i) {  // status of variable i (uniform i)
    return (int)((unsigned)__builtin_amdgcn_readlane((int)Sp, dl_of(i)) >> (4 * dk_of(i))) & 15;
}
__device__ __forceinline__ void st_set(unsigned &Sp, int i, int s) {
    const int sh = 4 * dk_of(i);
    if (lane_id() == dl_of(i)) Sp = (Sp & ~(15u << sh)) | ((unsigned)s << sh);
}
__device__ __forceinline__ int st_of(unsigned Sp, int k) { return (int)(Sp >> (4 * k)) & 15; }

// acc += w * column, column = N contiguous doubles (a column of V, a row of Ct, q); N even
__device__ __forceinline__ void axpy_dense(double2 (&acc)[NCH], const double *__restrict__ col, double w, int N) {
    const int lane = lane_id();
    double2 v[NCH];
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const int r = 2 * lane + 128 * m;
        v[m] = *reinterpret_cast<const double2 *>(col + (r < N ? r : 0));
    }
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const int r = 2 * lane + 128 * m;
        const double wk = (r < N) ? w : 0.0;
        acc[m].x = fma(v[m].x, wk, acc[m].x);
        acc[m].y = fma(v[m].y, wk, acc[m].y);
    }
}

// ------------------------------------------------------------------ state
struct Fac {
    double *L0;  // rows and columns < 64: packed columns, column c holds rows c..63 at cofs64(c) (slot c unused)
    double *L1;  // rows >= 64: element (r, c) at c * R1 + (r - 64)
    int R1;
    double *LR;  // big-factor build: the same rows once more BY ROWS, element (r, c) at (r - 64) * LR_STRIDE + c, zero for
                 // c >= r and in the rows >= K (what the back substitution streams)
};
__device__ __forceinline__ int cofs64(int c) { return c * 64 - ((c * (c - 1)) >> 1); }

struct Rows {  // one entry per row of the kept factor = per free variable, row r in lane r & 63 of slot r >> 6
    int ord[NSL], rank[NSL];           // variable index; its rank among the free variables by index (findall order)
    double zF[NSL], ur[NSL], dr[NSL];  // z, upper and lower bound of the variable
    double dg[NSL], rd[NSL];           // pivot d_r of the LDL' factor and its reciprocal
    double Y[NR][NSL];                 // forward-substituted border: Y[w] = (L^-1 [A;G][w,F]')_r, Y[CC] = (L^-1 c)_r
};

struct WLds {
    Fac F;
    // (the columns of [A;G] of the free variables, X[w]_r = [A;G][w, ord_r], are NOT kept in registers: the hot path
    //  never needs them -- the ratio test works from H and gz, the Gram matrix and gz take the column of a variable
    //  when it is appended or deleted -- and the rare paths (rank filter run in full, purged-row least squares,
    //  polishSz!) gather them from Ct)
    double *H;      // NR x NR, symmetric, full: H[a][b] = sum_r Y[a]_r Y[b]_r / d_r
    double *tr;     // scratch of small_spd_solve
    double *aLrow;  // alphaL by row id
    double *yn;     // 16 doubles: a new border row / the downdate vector
    double *GG;     // MJX x MJX, symmetric, full: Gram matrix of the rows of [A;G][:, F]: GG[a][b] = sum_r X[a]_r X[b]_r
    double *xn;     // 16 doubles: a column of [A;G] (for the rank-1 change of GG)
    int16_t *ra;    // kept row ids in order
    // big-factor build: a ring of RING_BYTES in LDS that LDS-DMA loads stream columns into (see "column streams" below)
    const double *ring;
    unsigned ringAddr;  // its LDS byte address (the M0 base of the DMA destination)
    int zidx;           // index (relative to F.L0) of an LDS double that is always zero
    const double *zero; // the same word as a pointer (every build has one)
};

// ------------------------------------------------------------------ column streams through an LDS ring (big-factor build)
// With one wavefront per SIMD nobody hides a wavefront's memory latency for it, and with 200+ free variables a pass
// streams several hundred KB (the factor twice, the free columns of V once): a loop that waits for four columns, uses
// them and asks for the next four spends its time waiting.  The three streaming loops of the big-factor build therefore
// keep ~14 KB in flight per wavefront in an LDS ring filled by LDS-DMA loads (global_load_lds_dwordx4: no register
// destination, so the depth costs no registers): slot s holds one column in pieces of 1 KiB (one wave-instruction: 64
// lanes x 16 B, destination = M0 base + lane * 16), columns are requested RING_SLOTS ahead and consumed in order behind a
// COUNTED s_waitcnt vmcnt (loads return in order; every slot takes exactly NI DMA instructions, so "at most NI * (D - 1)
// outstanding" means the oldest column has landed -- younger compiler-issued memory operations can only lengthen the wait).
// 1,024 wavefronts streaming this way reach 5.9 TB/s (tools/probes/glds_ring_probe.hip).
constexpr int RING_BYTES = 16 * 1024;

// One LDS-DMA load, 16 B per lane: global address = the UNIFORM 64-bit base `sbase` + this lane's byte offset `voff`;
// LDS destination = M0 base `lds_dst` + lane * 16.  The base travels in a scalar register pair and the per-lane offsets
// of a stream are loop-invariant, so a piece costs s_mov m0 + s_nop + the load.  M0 is NOT saved, and it cannot be
// declared clobbered either (LLVM treats m0 as a reserved register and ignores -- with a warning -- a clobber of it):
// nothing else in these kernels uses it (no ds_gws, no s_movrel, no LDS-direct), and tests/test_capi_cpu.py checks the
// built code objects for exactly that; the test fails, not skips, where the library was built.
__device__ __forceinline__ const double *uni_ptr(const double *p) {  // a uniform pointer, pinned to a scalar register pair
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return reinterpret_cast<const double *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void glds16_s(const void *sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0"
                 :
                 : "s"(sbase), "v"(voff), "s"(lds_dst)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Everything in flight has landed -- the prologue columns of a ring AND every load the compiler issued before: said with
// the BUILTIN, which the compiler's own wait-count bookkeeping understands (it cannot see through inline asm): without
// it the first use of an earlier load's result inside a ring loop gets a compiler-placed vmcnt(0) that executes in every
// iteration and drains the ring each time.
__device__ __forceinline__ void wait_all_landed() {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
    asm volatile("" ::: "memory");
}

// N contiguous doubles (a column of V, a row of Ct) into slot `slot`: piece m = elements 128 m + 2 lane, + 1.
// Every slot takes exactly NI DMA instructions whatever N is (the counted waits rely on it); every lane loads (lanes
// beyond N re-read element 0 -- no exec juggling per DMA).  The per-lane offsets are formed once per stream.
template <int NI>
__device__ __forceinline__ void dense_offsets(int N, unsigned (&vo)[NI]) {
    const int lane = lane_id();
#pragma unroll
    for (int m = 0; m < NI; ++m) {
        const int e = 128 * m + 2 * lane;
        vo[m] = (unsigned)(e < N ? e : 0) * 8u;
    }
}
template <int NI>
__device__ __forceinline__ void ring_issue_dense(const WLds &L, int slot, const double *colin, const unsigned (&vo)[NI]) {
    const double *col = uni_ptr(colin);
#pragma unroll
    for (int m = 0; m < NI; ++m) glds16_s(col, vo[m], L.ringAddr + (unsigned)(slot * NI + m) * 1024u);
}
// The part of column c of the factor that lives in global memory (rows 64 .. K - 1, only rows > c) into slot `slot`:
// element (r, c) lands at double index r - 64 of the slot.  The big-factor build keeps every element of that storage
// that is NOT an entry of the current factor -- rows >= K, rows <= c, columns >= K -- at ZERO (zeroed when a QP starts;
// an append writes row K of the columns < K, a delete clears the row it frees), so a lane whose two rows are not needed
// is pointed at rows 254, 255 of the same column (K <= 252: zero for ever), the slot holds zeros wherever the column
// has no entry and the reader needs no mask.  TRI = false: the caller knows c < 64 (every row >= 64 lies below).
constexpr unsigned FAC_ZERO_OFF = 190u * 8u;
constexpr int LR_STRIDE = 256;
constexpr unsigned ROW_ZERO_OFF = 254u * 8u;  // (row r holds zeros from column r on; r <= 251)
template <int NI>
__device__ __forceinline__ void faccol_offsets(int K, unsigned (&vo)[NI]) {
    const int lane = lane_id();
#pragma unroll
    for (int m = 0; m < NI; ++m) {
        const int r = 64 + 128 * m + 2 * lane;  // this lane's two rows: r, r + 1
        vo[m] = (r < K) ? (unsigned)(r - 64) * 8u : FAC_ZERO_OFF;
    }
}
template <int NI, bool TRI>
__device__ __forceinline__ void ring_issue_faccol(const WLds &L, int slot, int c, const unsigned (&vo)[NI]) {
    const int lane = lane_id();
    const double *src = uni_ptr(L.F.L1 + (size_t)c * L.F.R1);
#pragma unroll
    for (int m = 0; m < NI; ++m) {
        unsigned off = vo[m];
        if (TRI) off = (64 + 128 * m + 2 * lane + 1 > c) ? off : FAC_ZERO_OFF;
        glds16_s(src, off, L.ringAddr + (unsigned)(slot * NI + m) * 1024u);
    }
}

// [A;G][w, ord_r] for this lane's rows (a gather from the contiguous row w of Ct; cold paths only)
template <int SL>
__device__ __forceinline__ void gather_X(const double *__restrict__ Ct, int N, int w, const int (&ord)[NSL], int K,
                                         double (&x)[NSL]) {
    const int lane = lane_id();
#pragma unroll
    for (int t = 0; t < NSL; ++t) x[t] = 0.0;
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + KSLOT * t;
        const double v = Ct[(size_t)w * N + (r < K ? ord[t] : 0)];
        x[t] = (r < K) ? v : 0.0;
    }
}

// one column of the factor for this lane's rows, zero where there is no entry (rows <= c, rows >= K, !valid).
// Slots that lie wholly above the diagonal (every row <= c) are not read at all (uniform branch).
template <int SL>
__device__ __forceinline__ void load_col(const Fac &F, int K, int c, double (&l)[NSL]) {
    const int lane = lane_id();
    const bool valid = c < K;
    const int cc = valid ? c : 0;
#pragma unroll
    for (int t = 0; t < NSL; ++t) l[t] = 0.0;
    if (SL <= 2 || cc < 63) {
        const int c0 = cc < 63 ? cc : 63;
        const double v = F.L0[cofs64(c0) - c0 + (lane > c0 ? lane : c0)];
        l[0] = (valid && lane > cc && lane < K) ? v : 0.0;
    }
#pragma unroll
    for (int t = 1; t < SL; ++t) {
        if (SL <= 2 || KSLOT * t + 63 > cc) {  // (uniform)
            const int rr = lane + KSLOT * (t - 1);
            const int rl = rr < F.R1 ? rr : F.R1 - 1;
            const double v = F.L1[cc * F.R1 + rl];
            l[t] = (valid && KSLOT * t + lane > cc && KSLOT * t + lane < K) ? v : 0.0;
        }
    }
}
template <int SL>
__device__ __forceinline__ void load_cols4(const Fac &F, int K, int c0, double (&l)[4][NSL]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) load_col<SL>(F, K, c0 + u, l[u]);
}
// element (r, c), r > c, per-lane r (r may be in any slot), uniform c
template <int SL>
__device__ __forceinline__ double fac_get(const Fac &F, int r, int c) {
    const int c0 = c < 63 ? c : 63;
    const int r0 = r < 64 ? (r > c0 ? r : c0) : 63;
    const double a = F.L0[cofs64(c0) - c0 + r0];
    if (SL == 1) return a;
    int r1 = r >= 64 ? r - 64 : 0;
    r1 = r1 < F.R1 ? r1 : F.R1 - 1;
    const double b = F.L1[c * F.R1 + r1];
    return (r < 64) ? a : b;
}
template <int SL>
__device__ __forceinline__ void fac_put(const Fac &F, int r, int c, double v, bool on) {
    if (SL == 1) {
        if (on) F.L0[cofs64(c) - c + r] = v;
    } else {
        if (on && r < 64) F.L0[cofs64(c) - c + r] = v;
        if (on && r >= 64) F.L1[c * F.R1 + (r - 64)] = v;
    }
}

// ------------------------------------------------------------------ kept factor: append / delete
// column c of the factor for this lane's rows from L0 and the ring slot the column was streamed into (it has landed).
// TB = the slot the caller's columns lie in (c >> 6 >= TB): slots below TB hold no row > c and are not read; the ring
// part needs no mask (zeros where the column has no entry, see ring_issue_faccol), the L0 part reads a zero word of LDS
// where lane <= c -- no select on the DATA, so the reads of a group of columns go out back to back, straight into the
// registers that keep them, and are waited for once.
template <int SL, int NI, int TB>
__device__ __forceinline__ void ring_read_faccol(const WLds &L, int K, int c, double (&l)[NSL], int slotIdx = -1) {
    constexpr int D = RING_BYTES / (1024 * NI);
    const int lane = lane_id();
    if (slotIdx < 0) slotIdx = c % D;
    if (TB == 0) {
        const int c0 = c < 63 ? c : 63;
        l[0] = L.F.L0[(lane > c && lane < K) ? cofs64(c0) - c0 + lane : L.zidx];  // (rows >= K of L0 are stale)
    }
    const double *__restrict__ slot = L.ring + (size_t)slotIdx * (128 * NI);
#pragma unroll
    for (int t = (TB > 1 ? TB : 1); t < SL; ++t)
        if (KSLOT * (t - 1) < 128 * NI) l[t] = slot[KSLOT * (t - 1) + lane];  // (compile-time: the slot holds rows 64 .. 64 + 128 NI - 1)
}

// forward substitution y <- L^-1 y with the columns of the factor streamed through the ring (big-factor build, K > 63).
// NI = 1: K <= 192 (rows 64..191: one piece per column), NI = 2: up to 256 rows.  Four columns per round: the next
// round's entries come out of LDS together (one wait) while this round's are applied.  The loop is written out once per
// register slot the columns lie in (64 is a multiple of 4: a group's rows sit in ONE slot), so that which slots a group
// reads and updates is known at compile time.  Columns K .. 4 ng - 1 are all zero: applied like the others.
// ap(IC<tb>, l, lc): apply column (64 tb + l) of the factor, whose entries for this lane's rows are lc, to the right-hand
// side(s) -- y_c is final when its column comes up.
template <int SL, int NI, class AP>
__device__ __forceinline__ void fwd_sweep_ring(const WLds &L, int Kin, AP &&ap) {
    constexpr int D = RING_BYTES / (1024 * NI);
    static_assert(D >= 8 && D % 4 == 0 && D <= 64, "two groups of four columns in the ring; the prologue stays below column 64");
    const int K = uni(Kin);
    const int ng = (K + 3) >> 2;  // groups of four columns
    unsigned vo[NI];
    faccol_offsets<NI>(K, vo);
    for (int c = 0; c < D && c < 4 * ng; ++c) ring_issue_faccol<NI, false>(L, c, c, vo);
    double lc[4][NSL], ln[4][NSL];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < NSL; ++t) lc[u][t] = ln[u][t] = 0.0;
    wait_all_landed();
#pragma unroll
    for (int u = 0; u < 4; ++u) ring_read_faccol<SL, NI, 0>(L, K, u, lc[u]);
    sfor<0, SL>(SFOR_BODY(tb) {
        SFOR_IDX(tb);
        const int gEnd = (16 * (tb + 1) < ng) ? 16 * (tb + 1) : ng;
        // one round: group g sits in `cur` (read a round ago, the reads have completed); its ring slots take the group
        // D / 4 ahead, the next group's entries come out of LDS into `nxt` while this group is applied
        auto round = [&](int g, double (&cur)[4][NSL], double (&nxt)[4][NSL]) __attribute__((always_inline)) {
            const int c0 = 4 * g;
            const bool more = c0 + D < 4 * ng;
            if (more) {
#pragma unroll
                for (int u = 0; u < 4; ++u) ring_issue_faccol<NI, true>(L, (c0 + u) % D, c0 + D + u, vo);
            }
            if (g + 1 < ng) {
                if (more) wait_vm<NI * (D - 4)>();  // (groups g + 2 .. g + D / 4 stay in flight)
                else wait_vm<0>();
#pragma unroll
                for (int u = 0; u < 4; ++u) ring_read_faccol<SL, NI, tb>(L, K, c0 + 4 + u, nxt[u]);
            }
            const int l0 = c0 & 63;
#pragma unroll
            for (int u = 0; u < 4; ++u) ap(IC<tb>{}, l0 + u, cur[u]);
            wait_lds();  // (nxt has arrived: the slots of group g + 1 are free for the next round's request)
        };
        // (two rounds per trip, the two register sets changing roles: no copy between rounds)
        for (int g = 16 * tb; g < gEnd; g += 2) {
            round(g, lc, ln);
            if (g + 1 < gEnd) {
                round(g + 1, ln, lc);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < NSL; ++t) lc[u][t] = ln[u][t];
            }
        }
    });
}

// Append variable j as row K (SSQP.jl:322 refactors V[F,F] from scratch; here one forward substitution).
// lnew = the new row of L (for the border update).  Returns the new pivot (must be > 0).
template <int SL>
__device__ __forceinline__ double append_row(const WLds &L, Rows &R, int K, int j, const double *__restrict__ V, int N,
                                             double (&lnew)[NSL], double (&vraw)[NSL], double (&ysub)[NSL]) {
    const Fac &F = L.F;
    const int lane = lane_id();
    const double *__restrict__ col = V + (size_t)j * N;
    const double vjj = col[j];
    double y[NSL];
#pragma unroll
    for (int t = 0; t < NSL; ++t) y[t] = 0.0;
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + KSLOT * t;
        const double v = col[r < K ? R.ord[t] : j];
        y[t] = (r < K) ? v : 0.0;
    }
#pragma unroll
    for (int t = 0; t < NSL; ++t) vraw[t] = y[t];
    if (NSL > 2 && SL >= 2) {
        auto ap = [&](auto tbc, int l, const double (&lc)[NSL]) __attribute__((always_inline)) {
            constexpr int tb = decltype(tbc)::value;
            const double yc = readlane_f64(y[tb], l);
#pragma unroll
            for (int t = tb; t < SL; ++t) y[t] = fma(-lc[t], yc, y[t]);  // (slots above hold rows <= c: no entry)
        };
        if (K <= 192) fwd_sweep_ring<SL, 1>(L, K, ap);
        else fwd_sweep_ring<SL, 2>(L, K, ap);
    } else {
        double lc[4][NSL], ln[4][NSL];
        load_cols4<SL>(F, K, 0, lc);
        for (int c0 = 0; c0 < K; c0 += 4) {
            load_cols4<SL>(F, K, c0 + 4, ln);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (c0 + u < K) {  // uniform
                    const double yc = rbcast<SL>(y, c0 + u);
#pragma unroll
                    for (int t = 0; t < SL; ++t) y[t] = fma(-lc[u][t], yc, y[t]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < SL; ++t) lc[u][t] = ln[u][t];
        }
    }
#pragma unroll
    for (int t = 0; t < NSL; ++t) ysub[t] = y[t];
    double part = 0.0;
#pragma unroll
    for (int t = 0; t < NSL; ++t) lnew[t] = 0.0;
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + KSLOT * t;
        const double tr = (r < K) ? y[t] * R.rd[t] : 0.0;
        part = fma(y[t], tr, part);
        lnew[t] = tr;
        if (K < 64) {
            if (t == 0 && r < K) F.L0[cofs64(r) - r + K] = tr;
        } else if (SL >= 2) {
            if (r < K) F.L1[r * F.R1 + (K - 64)] = tr;
        }
    }
    if (NSL > 2 && SL >= 2 && K >= 64) {  // the same row in the by-rows copy, zeros from column K on
#pragma unroll
        for (int t = 0; t < NSL; ++t) F.LR[(size_t)(K - 64) * LR_STRIDE + lane + KSLOT * t] = lnew[t];
    }
    const double dnew = vjj - wave_sum(part);
    wave_sync();
    return dnew;
}

// Delete row p: L33 D3 L33' += d_p l l' on the trailing block (rank-1 update, one column per step), then the
// row/column is removed from the packed storage.  pv receives p_k = (L33^-1 l)_k in row k (the downdate of H
// and nothing else needs it); dg/rd of the rows > p are updated in place (still in their OLD positions).
// MERGE (one row slot, the whole factor in LDS): the updated column k is written straight to where the compaction would
// move it -- column k - 1, rows r - 1; that storage (the old column k - 1) was consumed an iteration ago -- and
// delete_compact only has the columns left of p to do.
template <int SL, bool MERGE = false>
__device__ __forceinline__ void delete_update(const Fac &F, Rows &R, int K, int p, double (&pv)[NSL], double (&bt)[NSL]) {
    static_assert(!MERGE || SL == 1, "the merged form is for the one-slot factor");
    const int lane = lane_id();
    double w[NSL];
    load_col<SL>(F, K, p, w);
#pragma unroll
    for (int t = 0; t < NSL; ++t) pv[t] = bt[t] = 0.0;
    double alpha = rbcast<SL>(R.dg, p);
    double lc[NSL], ln[NSL];
    load_col<SL>(F, K, p + 1, lc);
    for (int k = p + 1; k < K; ++k) {
        load_col<SL>(F, K, k + 1, ln);
        const double pk = rbcast<SL>(w, k);
        const double dk = rbcast<SL>(R.dg, k);
        const double dn = fma(alpha * pk, pk, dk);
        const double rdn = fast_rcp(dn);
        const double beta = pk * alpha * rdn;
        alpha = alpha * dk * rdn;
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int r = lane + KSLOT * t;
            if (r == k) {
                R.dg[t] = dn;
                R.rd[t] = rdn;
                pv[t] = pk;
                bt[t] = beta;
            }
            const bool below = r > k && r < K;
            w[t] = below ? fma(-pk, lc[t], w[t]) : w[t];
            if (MERGE) {
                if (below) F.L0[cofs64(k - 1) - (k - 1) + (r - 1)] = fma(beta, w[t], lc[t]);
            } else {
                fac_put<SL>(F, r, k, fma(beta, w[t], lc[t]), below);
            }
        }
#pragma unroll
        for (int t = 0; t < SL; ++t) lc[t] = ln[t];
    }
    wave_sync();
}
template <int SL, bool MERGE = false>
__device__ __forceinline__ void delete_compact(const Fac &F, int K, int p) {
    const int lane = lane_id();
    constexpr int CB = 4;  // columns per read / write round (their storage is disjoint)
    if (MERGE && SSQP_WAVE_VARIANT != 1) {  // (not in the eight-per-CU build: it costs that build nine more spilled registers and 3 %)
        // one row slot, packed columns in LDS: rows r > p of a column c < p move up by one INSIDE the column -- lane r reads
        // its lower neighbour's entry and stores it as its own; eight columns per round, reads before stores (a
        // wavefront's LDS operations execute in order)
        const bool mv = lane >= p && lane + 1 < K;
        const int src = lane + 1 < 64 ? lane + 1 : 63;
        for (int c0 = 0; c0 < p; c0 += 8) {
            double a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = (c0 + u < p) ? c0 + u : c0;
                a[u] = F.L0[cofs64(c) - c + src];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u;
                if (c < p && mv) F.L0[cofs64(c) - c + lane] = a[u];
            }
        }
        wave_sync();
        return;
    }
    for (int c0 = 0; c0 < p; c0 += CB) {  // columns c < p lose row p: rows r > p move up by one inside the column
        double a[CB][NSL];
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const int c = (c0 + u < p) ? c0 + u : c0;
#pragma unroll
            for (int t = 0; t < SL; ++t) {
                const int rn = lane + KSLOT * t;  // new row
                a[u][t] = (rn >= p && rn + 1 < K) ? fac_get<SL>(F, rn + 1, c) : 0.0;
            }
        }
        wave_sync();
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            if (c0 + u < p) {  // uniform
#pragma unroll
                for (int t = 0; t < SL; ++t) {
                    const int rn = lane + KSLOT * t;
                    fac_put<SL>(F, rn, c0 + u, a[u][t], rn >= p && rn + 1 < K);
                }
            }
        }
    }
    // column c -> c - 1, rows r > c -> r - 1, ascending: the target of column c is the storage of column c - 1, which
    // this round (or an earlier one) has already read  (MERGE: delete_update has written them in place)
    for (int c0 = p + 1; c0 < K && !MERGE; c0 += CB) {
        double a[CB][NSL];
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const int c = (c0 + u < K) ? c0 + u : c0;
#pragma unroll
            for (int t = 0; t < SL; ++t) {
                const int rn = lane + KSLOT * t;
                a[u][t] = (rn >= c && rn + 1 < K) ? fac_get<SL>(F, rn + 1, c) : 0.0;
            }
        }
        wave_sync();
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            if (c0 + u < K) {  // uniform
#pragma unroll
                for (int t = 0; t < SL; ++t) {
                    const int rn = lane + KSLOT * t;
                    fac_put<SL>(F, rn, c0 + u - 1, a[u][t], rn >= c0 + u && rn + 1 < K);
                }
            }
        }
        wave_sync();
    }
    wave_sync();
}

// The big-factor build's delete: delete_update and delete_compact in ONE pass over the factor, the global part of every
// column streamed through the ring.  Columns k > p are updated (the rank-1 recurrence above) and written straight to
// where the compaction would move them -- column k - 1, rows r - 1 (that storage has been consumed: the stream runs
// ahead of the writes); columns c < p only lose row p (their rows > p move up by one).  Both copies of the rows >= 64 are
// maintained (by columns: coalesced; by rows: one scattered store per slot and column), and the row the factor loses is
// zeroed in both (ring_issue_faccol / ring_issue_facrow rely on zeros outside the factor).
// Stores and DMA loads share the vmcnt counter and complete out of order with respect to each other, so a counted wait
// cannot tell "the oldest column has landed" here: the ring is used as two halves of four columns -- the next group is
// requested, this group is processed (with its stores), then vmcnt(0).
template <int SL, int NI>
__device__ __forceinline__ void delete_stream(const WLds &L, Rows &R, int Kin, int pin, double (&pv)[NSL], double (&bt)[NSL]) {
    const Fac &F = L.F;
    const int lane = lane_id();
    const int K = uni(Kin), p = uni(pin);
    double w[NSL];
    load_col<SL>(F, K, p, w);  // column p (rows > p), before its storage is overwritten
#pragma unroll
    for (int t = 0; t < NSL; ++t) pv[t] = bt[t] = 0.0;
    unsigned vo[NI];
    faccol_offsets<NI>(K, vo);
    double lc[4][NSL];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < NSL; ++t) lc[u][t] = 0.0;
    auto put = [&](int rn, int cn, double val, bool on) __attribute__((always_inline)) {  // element (rn, cn) of the new factor
        if (on && rn >= 64) {
            F.L1[cn * F.R1 + (rn - 64)] = val;
            F.LR[(size_t)(rn - 64) * LR_STRIDE + cn] = val;
        }
        if (on && rn < 64) F.L0[cofs64(cn) - cn + rn] = val;  // (rn < 64: cn < rn < 64)
    };
    wait_all_landed();
    // ---- columns c < p: rows r > p move up by one
    if (p > 0 && p < K - 1) {  // uniform
#pragma unroll
        for (int u = 0; u < 4; ++u) ring_issue_faccol<NI, true>(L, u, u, vo);
        wait_vm<0>();
        for (int j = 0, c0 = 0; c0 < p; ++j, c0 += 4) {
            const int sb = (j & 1) * 4;
            if (c0 + 4 < p) {
#pragma unroll
                for (int u = 0; u < 4; ++u) ring_issue_faccol<NI, true>(L, (4 - sb) + u, c0 + 4 + u, vo);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) ring_read_faccol<SL, NI, 0>(L, K, c0 + u, lc[u], sb + u);
            wait_lds();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (c0 + u < p) {  // uniform
                    shift_up<SL>(lc[u], p);  // lane of row rn >= p now holds the old row rn + 1
#pragma unroll
                    for (int t = 0; t < SL; ++t) {
                        const int rn = lane + KSLOT * t;
                        put(rn, c0 + u, lc[u][t], rn >= p && rn + 1 < K);
                    }
                }
            }
            wait_vm<0>();
        }
    }
    // ---- columns k > p: the rank-1 update, written to column k - 1, rows r - 1
    double alpha = rbcast<SL>(R.dg, p);
    if (p + 1 < K) {  // uniform
#pragma unroll
        for (int u = 0; u < 4; ++u) ring_issue_faccol<NI, true>(L, u, p + 1 + u, vo);
        wait_vm<0>();
        for (int j = 0, k0 = p + 1; k0 < K; ++j, k0 += 4) {
            const int sb = (j & 1) * 4;
            if (k0 + 4 < K) {
#pragma unroll
                for (int u = 0; u < 4; ++u) ring_issue_faccol<NI, true>(L, (4 - sb) + u, k0 + 4 + u, vo);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) ring_read_faccol<SL, NI, 0>(L, K, k0 + u, lc[u], sb + u);
            wait_lds();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u;
                if (k < K) {  // uniform
                    const double pk = rbcast<SL>(w, k);
                    const double dk = rbcast<SL>(R.dg, k);
                    const double dn = fma(alpha * pk, pk, dk);
                    const double rdn = fast_rcp(dn);
                    const double beta = pk * alpha * rdn;
                    alpha = alpha * dk * rdn;
#pragma unroll
                    for (int t = 0; t < SL; ++t) {
                        const int r = lane + KSLOT * t;
                        if (r == k) {
                            R.dg[t] = dn;
                            R.rd[t] = rdn;
                            pv[t] = pk;
                            bt[t] = beta;
                        }
                        const bool below = r > k && r < K;
                        w[t] = below ? fma(-pk, lc[u][t], w[t]) : w[t];
                        put(r - 1, k - 1, fma(beta, w[t], lc[u][t]), below);
                    }
                }
            }
            wait_vm<0>();
        }
    }
    // ---- the row the factor loses: zero in both copies
    if (K - 1 >= 64) {
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int c = lane + KSLOT * t;
            if (c < K - 1) F.L1[c * F.R1 + (K - 1 - 64)] = 0.0;
        }
#pragma unroll
        for (int t = 0; t < NSL; ++t) F.LR[(size_t)(K - 1 - 64) * LR_STRIDE + lane + KSLOT * t] = 0.0;
    }
    wave_sync();
}

// forward substitution L y = b of the border columns selected by `cols` (bit w), from their raw right-hand sides
template <int SL>
__device__ __forceinline__ void border_sweep(const Fac &F, Rows &R, int K, unsigned cols, const double *__restrict__ Ct,
                                             int N, const double2 (&hq)[NCH], const WLds *Lw = nullptr) {
    const int lane = lane_id();
#pragma unroll
    for (int w = 0; w < NR; ++w) {
        if ((cols >> w) & 1u) {  // uniform
            if (w == CC) {
#pragma unroll
                for (int t = 0; t < SL; ++t) {
                    const int r = lane + KSLOT * t;
                    const double v = dense_gather(hq, (r < K) ? R.ord[t] : 0);  // c = hq[F]  (SSQP.jl:324)
                    R.Y[w][t] = (r < K) ? v : 0.0;
                }
            } else {
                double x[NSL];
                gather_X<SL>(Ct, N, w < MJX ? w : 0, R.ord, K, x);
#pragma unroll
                for (int t = 0; t < SL; ++t) R.Y[w][t] = x[t];
            }
        }
    }
    if (NSL > 2 && SL >= 2) {  // (big-factor build: the factor's columns streamed through the ring)
        auto ap = [&](auto tbc, int l, const double (&lcol)[NSL]) __attribute__((always_inline)) {
            constexpr int tb = decltype(tbc)::value;
#pragma unroll
            for (int w = 0; w < NR; ++w) {
                if ((cols >> w) & 1u) {  // uniform
                    const double yc = readlane_f64(R.Y[w][tb], l);
#pragma unroll
                    for (int t = tb; t < SL; ++t) R.Y[w][t] = fma(-lcol[t], yc, R.Y[w][t]);
                }
            }
        };
        if (K <= 192) fwd_sweep_ring<SL, 1>(*Lw, K, ap);
        else fwd_sweep_ring<SL, 2>(*Lw, K, ap);
        return;
    }
    double lc[4][NSL], ln[4][NSL];
    load_cols4<SL>(F, K, 0, lc);
    for (int c0 = 0; c0 < K; c0 += 4) {
        load_cols4<SL>(F, K, c0 + 4, ln);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c0 + u < K) {  // uniform
#pragma unroll
                for (int w = 0; w < NR; ++w) {
                    if ((cols >> w) & 1u) {
                        const double yc = rbcast<SL>(R.Y[w], c0 + u);
#pragma unroll
                        for (int t = 0; t < SL; ++t) R.Y[w][t] = fma(-lc[u][t], yc, R.Y[w][t]);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < SL; ++t) lc[u][t] = ln[u][t];
    }
}

// back substitution L' x = v (unit upper), v in the row registers
// L(r, c) for uniform r and c = this lane's row index (a gathered row of the factor); 0 where c >= r
template <int SL>
__device__ __forceinline__ void load_rowT(const Fac &F, int r, double (&l)[NSL]) {
    const int lane = lane_id();
#pragma unroll
    for (int t = 0; t < NSL; ++t) l[t] = 0.0;
    if (SL == 1) {  // (no branch: rows <= 0 have no live lane)
        const bool live = lane < r;
        const double x = F.L0[live ? cofs64(lane) - lane + r : 0];
        l[0] = live ? x : 0.0;
        return;
    }
    if (r <= 0) return;  // uniform
    if (r < 64) {
        const bool live = lane < r;
        const double x = F.L0[live ? cofs64(lane) - lane + r : 0];
        l[0] = live ? x : 0.0;
    } else {
#pragma unroll
        for (int t = 0; t < SL; ++t) {  // column lane + 64 t of row r (columns < 64 all lie left of row r)
            const bool live = KSLOT * t + lane < r;
            const double b = F.L1[(live ? KSLOT * t + lane : 0) * F.R1 + (r - 64)];
            l[t] = live ? b : 0.0;
        }
    }
}
template <int SL>
__device__ __forceinline__ void back_sweep(const Fac &F, int K, double (&v)[NSL]) {
    double lc[4][NSL], ln[4][NSL];
#pragma unroll
    for (int u = 0; u < 4; ++u) load_rowT<SL>(F, K - 1 - u, lc[u]);
    for (int r0 = K - 1; r0 > 0; r0 -= 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) load_rowT<SL>(F, r0 - 4 - u, ln[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r0 - u > 0) {  // uniform
                const double xr = rbcast<SL>(v, r0 - u);
#pragma unroll
                for (int t = 0; t < SL; ++t) v[t] = fma(-lc[u][t], xr, v[t]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < NSL; ++t) lc[u][t] = ln[u][t];
    }
}

// The back substitution of the big-factor build goes by ROWS: once x_r is final, v_c -= L(r, c) x_r for every c < r is
// an AXPY with row r of the factor -- no sum over lanes, no triangle to resolve inside a block.  Row r of the
// column-major storage would cost one cache line per lane, so the build keeps the rows >= 64 a second time by rows
// (Fac::LR: an append writes its row to both, a delete re-forms the rows it changed) and streams them through the ring
// from the last row down, four rows per round, the loop written out once per register slot the rows lie in.  Rows < 64
// come from the packed columns in LDS afterwards.  NI = 1: K <= 128 (a row has at most 128 entries), NI = 2: up to 256.
template <int NI>
__device__ __forceinline__ void ring_issue_facrow(const WLds &L, int slot, int r, const unsigned (&vo)[NI]) {
    const int lane = lane_id();
    const double *src = uni_ptr(L.F.LR + (size_t)(r - 64) * LR_STRIDE);
#pragma unroll
    for (int m = 0; m < NI; ++m) {
        const unsigned off = (128 * m + 2 * lane < r) ? vo[m] : ROW_ZERO_OFF;  // (this lane's columns: 128 m + 2 lane, + 1)
        glds16_s(src, off, L.ringAddr + (unsigned)(slot * NI + m) * 1024u);
    }
}
template <int NI, int TB>
__device__ __forceinline__ void ring_read_facrow(const WLds &L, int r, double (&l)[NSL]) {
    constexpr int D = RING_BYTES / (1024 * NI);
    const int lane = lane_id();
    const double *__restrict__ slot = L.ring + (size_t)(r % D) * (128 * NI);
#pragma unroll
    for (int t = 0; t <= TB; ++t)
        if (KSLOT * t < 128 * NI) l[t] = slot[KSLOT * t + lane];
}
template <int SL, int NI>
__device__ __forceinline__ void back_sweep_rows(const WLds &L, int Kin, double (&v)[NSL]) {
    constexpr int D = RING_BYTES / (1024 * NI), DB = D / 4;
    const int lane = lane_id();
    const int K = uni(Kin);
    if (K <= 1) return;
    if (K > 64) {
        const int nb = ((K - 1) >> 2) + 1;  // blocks of four rows; blocks 16 .. nb - 1 hold the rows >= 64 (rows >= K: zeros)
        unsigned vo[NI];
#pragma unroll
        for (int m = 0; m < NI; ++m) vo[m] = (unsigned)(128 * m + 2 * lane) * 8u;
        for (int b = nb - 1; b > nb - 1 - DB && b >= 16; --b)
#pragma unroll
            for (int u = 0; u < 4; ++u) ring_issue_facrow<NI>(L, (4 * b + u) % D, 4 * b + u, vo);
        double lc[4][NSL], ln[4][NSL];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < NSL; ++t) lc[u][t] = ln[u][t] = 0.0;
        wait_all_landed();
#pragma unroll
        for (int u = 0; u < 4; ++u) ring_read_facrow<NI, SL - 1>(L, 4 * (nb - 1) + u, lc[u]);
        wait_lds();
        sfor<0, SL - 1>(SFOR_BODY(tq) {
            SFOR_IDX(tq);
            constexpr int tb = SL - 1 - tq;  // SL - 1 .. 1: the slot the block's rows lie in
            const int bTop = (16 * (tb + 1) < nb ? 16 * (tb + 1) : nb) - 1;
            // one round: block b sits in `cur` (its reads have completed); its ring slots take the block DB below, the
            // block below comes out of LDS into `nxt` while this one is applied
            auto round = [&](int b, double (&cur)[4][NSL], double (&nxt)[4][NSL]) __attribute__((always_inline)) {
                const int c0 = 4 * b;
                if (b - DB >= 16) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) ring_issue_facrow<NI>(L, (c0 + u) % D, 4 * (b - DB) + u, vo);
                }
                if (b > 16) {
                    if (b - DB >= 16) wait_vm<NI * (D - 4)>();  // (blocks b - 2 .. b - DB stay in flight)
                    else wait_vm<0>();
#pragma unroll
                    for (int u = 0; u < 4; ++u) ring_read_facrow<NI, tb>(L, c0 - 4 + u, nxt[u]);
                }
                const int l0 = c0 & 63;
#pragma unroll
                for (int u = 3; u >= 0; --u) {
                    const double xr = readlane_f64(v[tb], l0 + u);  // (final: every row beyond has been applied)
#pragma unroll
                    for (int t = 0; t <= tb; ++t) v[t] = fma(-cur[u][t], xr, v[t]);  // (zeros from column r on)
                }
                wait_lds();
            };
            // (two rounds per trip, the two register sets changing roles: no copy between rounds)
            for (int b = bTop; b >= 16 * tb; b -= 2) {
                round(b, lc, ln);
                if (b - 1 >= 16 * tb) {
                    round(b - 1, ln, lc);
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int t = 0; t < NSL; ++t) lc[u][t] = ln[u][t];
                }
            }
        });
    }
    // rows 63 .. 1: a gathered row of the packed columns in LDS, four rows per round
    const int K0 = K < 64 ? K : 64;
    auto rowT = [&](int r) -> double {  // L(r, c) for c = lane < r, else 0
        const bool live = lane < r;
        const double x = L.F.L0[live ? cofs64(lane) - lane + r : L.zidx];
        return x;
    };
    double rc[4], rn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) rc[u] = rowT(K0 - 1 - u);
    for (int r0 = K0 - 1; r0 > 0; r0 -= 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) rn[u] = rowT(r0 - 4 - u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double xr = readlane_f64(v[0], (r0 - u) & 63);  // (rows <= 0: nothing live, the product is zero)
            v[0] = fma(-rc[u], xr, v[0]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) rc[u] = rn[u];
    }
}

// ------------------------------------------------------------------ getRowsGJr (utils.jl:49-86) in registers
// X = [AE bE]: lanes own the columns (free variable r -> lane r & 63 of slot r >> 6, the right-hand side is
// column K), the rows are the ACTIVE rows of [A;G] in increasing row id -- the row loop is unrolled over the row
// ids, inactive ones are skipped by a uniform branch, so every register index is static.  The reference's column
// permutation c0 is a position per column: initially the variable's rank among the free variables (findall
// order), K for the right-hand side; the pivot is the FIRST maximum of |X[i, c0[j:nc]]| in c0 order.  Arithmetic
// as in the reference, operation for operation (IEEE division, multiply and subtract rounded separately).
// Returns the kept rows as a bit mask over row ids.
template <int CS>
__device__ __forceinline__ unsigned rank_filter(const Rows &R, double bEv, unsigned act, int K, double tol,
                                                const double *__restrict__ Ct, int N) {
    const int lane = lane_id();
    const int nc = K + 1;
    double x[MJX][CS];
    int posn[CS];
    sfor<0, CS>(SFOR_BODY(cs) {
        SFOR_IDX(cs);
        const int t = lane + KSLOT * cs;
        posn[cs] = (t < K) ? R.rank[cs] : ((t == K) ? K : 0x7fff0000);
        sfor<0, MJX>(SFOR_BODY(i) {
            SFOR_IDX(i);
            const double be = readlane_f64(bEv, i);
            double xv = 0.0;
            if ((act >> i) & 1u) xv = Ct[(size_t)i * N + (t < K ? R.ord[cs] : 0)];  // uniform branch; [A;G][i, ord]
            x[i][cs] = (t < K) ? xv : ((t == K) ? be : 0.0);
        });
    });
    unsigned kept = 0;
    int j = 0;
    sfor<0, MJX>(SFOR_BODY(i) {
        SFOR_IDX(i);
        if (((act >> i) & 1u) && j < nc) {  // uniform
            double am = -1.0;
            int ap = 0x7fffffff;
            sfor<0, CS>(SFOR_BODY(cs) {
                SFOR_IDX(cs);
                const bool in = (posn[cs] >= j) & (posn[cs] < nc);
                const double a = in ? fabs(x[i][cs]) : -1.0;
                const bool better = (a > am) | ((a == am) & (posn[cs] < ap));
                am = better ? a : am;
                ap = better ? posn[cs] : ap;
            });
            const double m = wave_max(am);
            if (m > tol) {  // utils.jl:61 (uniform)
                unsigned long long tie = __ballot(am == m);
                int mlane = __ffsll((long long)tie) - 1;
                int mpos = __builtin_amdgcn_readlane(ap, mlane);
                tie &= tie - 1;
                while (tie) {
                    const int l2 = __ffsll((long long)tie) - 1;
                    const int p2 = __builtin_amdgcn_readlane(ap, l2);
                    mlane = p2 < mpos ? l2 : mlane;
                    mpos = p2 < mpos ? p2 : mpos;
                    tie &= tie - 1;
                }
                kept |= 1u << i;
                // c0[mj] <-> c0[j]; the pivot column broadcasts its entries of this row and the active rows below
                double dcol[MJX];
                sfor<0, MJX>(SFOR_BODY(k) {
                    SFOR_IDX(k);
                    dcol[k] = 0.0;
                });
                sfor<0, CS>(SFOR_BODY(cs) {
                    SFOR_IDX(cs);
                    const int pz = posn[cs];
                    posn[cs] = (pz == mpos) ? j : ((pz == j) ? mpos : pz);
                    const unsigned long long own = __ballot(posn[cs] == j);
                    if (own) {  // uniform: the owner lane holds the pivot column in this slot
                        const int src = __ffsll((long long)own) - 1;
                        sfor<i, MJX>(SFOR_BODY(k) {
                            SFOR_IDX(k);
                            if ((act >> k) & 1u) dcol[k] = readlane_f64(x[k][cs], src);
                        });
                    }
                });
                sfor<0, CS>(SFOR_BODY(cs) {
                    SFOR_IDX(cs);
                    const double xn = x[i][cs] / dcol[i];  // utils.jl:68-70
                    sfor<i + 1, MJX>(SFOR_BODY(k) {        // utils.jl:71-78 (rows above are never read again)
                        SFOR_IDX(k);
                        if ((act >> k) & 1u) x[k][cs] = sub_mul_nc(x[k][cs], dcol[k], xn);
                    });
                });
                j += 1;
            }
        }
    });
    return kept;
}

// ------------------------------------------------------------------ Schur block
// H[a][b] -= / += over all pairs, three entries per lane
__device__ __forceinline__ void h_rank1(const WLds &L, double scale) {  // H += scale * yn yn'
    const int lane = lane_id();
    // (all nine reads first, from clamped addresses: a read inside the "e < NR * NR" branch, or behind the previous
    //  entry's store, waits for its own LDS round trip)
    double ya[3], yb[3], h[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int e = lane + 64 * q, ee = e < NR * NR ? e : 0;
        const int a = ee / NR, b = ee - a * NR;
        ya[q] = L.yn[a];
        yb[q] = L.yn[b];
        h[q] = L.H[ee];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int e = lane + 64 * q;
        if (e < NR * NR) L.H[e] = fma(ya[q] * scale, yb[q], h[q]);
    }
}

// GG += sign * xn xn'  (two entries per lane)
__device__ __forceinline__ void gg_rank1(const WLds &L, double sign) {
    const int lane = lane_id();
    double xa[2], xb[2], g[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = lane + 64 * q, ee = e < MJX * MJX ? e : 0;
        const int a = ee / MJX, b = ee - a * MJX;
        xa[q] = L.xn[a];
        xb[q] = L.xn[b];
        g[q] = L.GG[ee];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = lane + 64 * q;
        if (e < MJX * MJX) L.GG[e] = fma(xa[q] * sign, xb[q], g[q]);
    }
}

// Can getRowsGJr purge a row of X = [AE bE] at all?  After the eliminations by the rows kept before it, row i of the
// filter is the one vector of x_i + span(previous rows) that vanishes in the pivot columns -- whatever columns the
// reference picks, its 2-norm is at least delta_i = dist(x_i, span(previous rows)), so its largest entry is at least
// delta_i / sqrt(K + 1).  The delta_i^2 are the pivots of the LDL' factorisation of the Gram matrix X X' of the
// active rows: when every pivot exceeds its threshold (orders of magnitude above both tol^2 (K+1) and the rounding of
// the kept Gram matrix) no row can be purged and the filter's result is "all rows kept" -- exactly, without running
// it.  The Gram matrix is taken over the columns of AE alone: the right-hand side column can only enlarge every
// distance, and without it the certificate depends on (E, F) only -- and its pivots can only GROW when a column is
// appended to F or a row leaves E, so a certificate stays valid across such passes (WState::certMask).
// Lane i holds row i of the W0 x W0 Gram matrix of the active rows (row-id order, ids in L.ra).
template <int WM>
__device__ __forceinline__ bool full_rank_certified(const WLds &L, int W0, int raLane) {
    const int lane = lane_id();
    const int ri = lane < W0 ? raLane : 0;  // raLane: L.ra[lane], the row id of position `lane`
    // (an entry that does not exist is read from a zero word of LDS: with a select on the DATA the compiler puts every
    //  read under its own exec mask and waits for each round trip; here the WM reads go out together)
    const int zG = (int)(L.zero - L.GG);
    double a[WM], dgg[WM];
#pragma unroll
    for (int c = 0; c < WM; ++c) {
        const int rc = __builtin_amdgcn_readlane(raLane, c < W0 ? c : 0);
        a[c] = L.GG[(lane < W0 && c < W0) ? ri * MJX + rc : zG];
        dgg[c] = L.GG[rc * MJX + rc];  // (the diagonal entry the threshold of step c is taken from)
    }
    bool ok = true;
#pragma unroll
    for (int c = 0; c < WM; ++c) {
        if (c < W0) {  // uniform
            const double d = readlane_f64(a[c], c);
            const double thr = fmax(1e-8, 1e-6 * dgg[c]);
            if (!(d > thr)) ok = false;
            const double r = fast_rcp(d > thr ? d : 1.0);
            const double lic = a[c] * r;
#pragma unroll
            for (int c2 = 0; c2 < WM; ++c2) {
                if (c2 > c) {
                    const double bq = readlane_f64(a[c], c2);
                    a[c2] = fma(-lic, bq, a[c2]);
                }
            }
        }
    }
    return ok;
}

// lambda of the Schur system H[kept,kept] lam = bE[kept] + t[kept] (W <= WM), one wavefront: lane i takes row i of
// the kept block straight from the kept 12 x 12 matrix (row ids in L.ra) -- no staging copy; eliminations broadcast
// the pivot row with v_readlane; the unit-lower factor goes to `tr` column by column so that the back substitution
// needs one broadcast per step.  Returns false when a pivot is not > 0 (cholesky(C) of the reference throws).
template <int WM>
__device__ __forceinline__ bool schur_solve(const WLds &L, double bEv, int W, double &lam, int raLane) {
    const int lane = lane_id();
    const int ri = lane < W ? raLane : 0;  // raLane: L.ra[lane]
    const int zH = (int)(L.zero - L.H);   // (a zero word of LDS for the entries that do not exist: see full_rank_certified)
    double a[WM];
#pragma unroll
    for (int c = 0; c < WM; ++c) {
        const int rc = __builtin_amdgcn_readlane(raLane, c < W ? c : 0);
        a[c] = L.H[(lane < W && c < W) ? ri * NR + rc : zH];
    }
    double y = bperm_f64(bEv, ri) + L.H[ri * NR + CC];
    y = (lane < W) ? y : 0.0;
    double *tr = L.tr;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < WM; ++c) {
        if (c < W) {  // uniform
            const double d = readlane_f64(a[c], c);
            if (!(d > 0.0)) ok = false;
            const double r = fast_rcp(d);
            const double yc = readlane_f64(y, c);
            const double lic = a[c] * r;
            const bool below = lane > c;
            y = below ? fma(-lic, yc, y) : ((lane == c) ? y * r : y);
#pragma unroll
            for (int c2 = 0; c2 < WM; ++c2) {
                if (c2 > c) {
                    const double bq = readlane_f64(a[c], c2);
                    a[c2] = fma(-lic, bq, a[c2]);
                }
            }
            if (below && lane < WM) tr[c * WM + lane] = lic;
        }
    }
    wave_sync();
    double u[WM];
#pragma unroll
    for (int i = 1; i < WM; ++i) u[i] = tr[(lane < WM ? lane : 0) * WM + i];
#pragma unroll
    for (int i = WM - 1; i >= 1; --i) {
        if (i < W) {  // uniform
            const double xi = readlane_f64(y, i);
            y = (lane < i) ? fma(-u[i], xi, y) : y;
        }
    }
    lam = y;
    return ok;
}

}  // namespace wv

using namespace wv;

// ====================================================================================================
// one QP, one wavefront
// ====================================================================================================
struct WCtx {
    int N, M, J, MJ;
    double tol, tolG;
    const double *__restrict__ V;
    const double *__restrict__ Ct;
    const double *__restrict__ rhs;
    const double *__restrict__ q;
    const double *__restrict__ dlo;
    const double *__restrict__ uhi;
    ssqp_trace *trace;
    int ntrace;
    double *lamOut, *gamOut;  // this QP's multiplier outputs (null: not requested)
    const float *V32;         // big-factor build: this QP's V rounded to fp32 (column i at V32 + i N), or null (N > 256)
    double vmax;              // max |V_ij| (what the screening's error bound needs)
    long long nScreen, nCand; // (statistics: screened passes, exact columns they asked for)
    int RC;
    long long iter, ret;
    int det;
    long long sBytes, sRead, sFlops, sK3;
    int maxK;
#ifdef SSQP_PHASE_PROFILE
    unsigned long long ph[16];
    unsigned pn[16];
#endif
};

enum { W_CONTINUE = 0, W_BREAK = 1, W_HANDOVER = 2 };

// every dense vector / status update that follows a change of z[j] of a BOUND variable by dz (it entered B at a
// nonzero value, or leaves B): hq += V[:,j] dz, bEall -= [A;G][:,j] dz  (the caches of SSQP.jl:295,324)
__device__ __forceinline__ void bound_shift(const WCtx &C, double2 (&hq)[NCH], double &bEv, int j, double dz) {
    const int lane = lane_id();
    const double cj = C.Ct[(size_t)(lane < C.MJ ? lane : 0) * C.N + j];  // (requested with the column: one round trip)
    axpy_dense(hq, C.V + (size_t)j * C.N, dz, C.N);
    bEv = (lane < C.MJ) ? fma(-cj, dz, bEv) : bEv;
}

template <int SL>
__device__ __forceinline__ void recompute_H_c(const WLds &L, const Rows &R, int K, int MJ) {
    // t = H[:, c]: H[a][c] = H[c][a] = sum_r Y[a]_r y_c,r / d_r for every row a of [A;G]
    const int lane = lane_id();
#pragma unroll
    for (int a = 0; a < MJX; ++a) {
        if (a < MJ) {  // uniform
            double s = 0.0;
#pragma unroll
            for (int t = 0; t < SL; ++t) {
                const int r = lane + KSLOT * t;
                s = (r < K) ? fma(R.Y[a][t] * R.rd[t], R.Y[CC][t], s) : s;
            }
            s = wave_sum(s);
            if (lane == 0) {
                L.H[a * NR + CC] = s;
                L.H[CC * NR + a] = s;
            }
        }
    }
    wave_sync();
}
template <int SL>
__device__ __forceinline__ void recompute_H_all(const WLds &L, const Rows &R, int K, int MJ) {
    const int lane = lane_id();
#pragma unroll
    for (int a = 0; a < NR; ++a) {
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            if ((a < MJ || a == CC) && (b < MJ || b == CC)) {  // uniform
                double s = 0.0;
#pragma unroll
                for (int t = 0; t < SL; ++t) {
                    const int r = lane + KSLOT * t;
                    s = (r < K) ? fma(R.Y[a][t] * R.rd[t], R.Y[b][t], s) : s;
                }
                s = wave_sum(s);
                if (lane == 0) {
                    L.H[a * NR + b] = s;
                    L.H[b * NR + a] = s;
                }
            }
        }
    }
    wave_sync();
}

// Append variable j: factor row, per-row registers, border row, Schur block.  Returns false when the new pivot is
// not > 0 (cholesky(V[F,F]) of the reference throws, SSQP.jl:322).
// dz != 0: the variable leaves B at a nonzero value, so c = hq[F] has just changed by dz * V[F, j] for the rows already
// in the factor (hq itself is up to date).  L^-1 V[F, j] is the vector this append substitutes anyway, so the border
// column y_c and t = H[:, c] follow in O(K) instead of a re-gather, a re-sweep and eleven sums:
//   y_c += dz * y,   H[w][c] += dz * sum_r Y[w]_r lnew_r  (the same sums the new border row needs).
template <int SL>
__device__ __forceinline__ bool append_var(WCtx &C, const WLds &L, Rows &R, int &K, int j, const double2 (&hq)[NCH],
                                           const double *zg, double dz, double &gz) {
    const int lane = lane_id();
    const int N = C.N, MJ = C.MJ;
    // everything the new row needs from memory is requested before the factor sweep: one round trip, hidden
    const double cj = C.Ct[(size_t)(lane < MJ ? lane : 0) * N + j];  // column j of [A;G], row w in lane w
    const double uj = C.uhi[j], dj = C.dlo[j];
    const double zj = __hip_atomic_load(zg + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (written by this wavefront earlier)
    double lnew[NSL], vraw[NSL], ysub[NSL];
    WPH_DECL;
    const double dnew = append_row<SL>(L, R, K, j, C.V, N, lnew, vraw, ysub);
    WPH(14);  // (inside "one append": the gather of V[F, j], the forward sweep, the new row's stores)
    if (!(dnew > 0.0)) return false;
    if (dz != 0.0) {  // uniform
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int r = lane + KSLOT * t;
            R.Y[CC][t] = (r < K) ? fma(dz, ysub[t], R.Y[CC][t]) : R.Y[CC][t];
        }
    }
    const double rdn = fast_rcp(dnew);
    // rank among the free variables by index: rows with a larger index move up by one
    int below = 0;
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + KSLOT * t;
        const bool lt = (r < K) && (R.ord[t] < j);
        below += __popcll(__ballot(lt));
        R.rank[t] = ((r < K) && (R.ord[t] > j)) ? R.rank[t] + 1 : R.rank[t];
    }
    const double hj = dense_get(hq, j);
    gz = (lane < MJ) ? fma(cj, zj, gz) : gz;  // [A;G][:, F] z_F gains the new variable's term
    set_row_i<SL>(R.ord, K, j);
    set_row_i<SL>(R.rank, K, below);
    set_row<SL>(R.zF, K, zj);
    set_row<SL>(R.ur, K, uj);
    set_row<SL>(R.dr, K, dj);
    set_row<SL>(R.dg, K, dnew);
    set_row<SL>(R.rd, K, rdn);
    // border row: y_K,w = x_w - sum_c L(K,c) y_c,w  (the twelve sums in one butterfly)
    double prod[NR], sums[NR];
#pragma unroll
    for (int w = 0; w < NR; ++w) {
        double s = 0.0;
        if (w < MJ || w == CC) {  // uniform
#pragma unroll
            for (int t = 0; t < SL; ++t) s = (lane + KSLOT * t < K) ? fma(lnew[t], R.Y[w][t], s) : s;
        }
        prod[w] = s;
    }
    wave_sum_multi<NR>(prod, sums);
    {   // lane w takes its own sum: the new border row goes to LDS (and t_w = H[w][c] follows the change of c) in ONE
        // parallel step instead of a chain of read-modify-writes by lane 0
        double sl = 0.0;
#pragma unroll
        for (int w = 0; w < NR; ++w) sl = (lane == w) ? sums[w] : sl;
        const bool live = lane < MJ || lane == CC;
        const double yl = ((lane == CC) ? hj : cj) - sl;
        if (dz != 0.0 && lane < MJ) {
            const double hv = fma(dz, sl, L.H[lane * NR + CC]);
            L.H[lane * NR + CC] = hv;
            L.H[CC * NR + lane] = hv;
        }
        if (lane < NR) L.yn[lane] = live ? yl : 0.0;
    }
#pragma unroll
    for (int w = 0; w < NR; ++w) {
        if (w < MJ || w == CC) {  // uniform
            const double xw = (w == CC) ? hj : readlane_f64(cj, w < MJX ? w : 0);
            set_row<SL>(R.Y[w], K, xw - sums[w]);
        }
    }
    if (lane < 16) L.xn[lane] = (lane < MJ) ? cj : 0.0;
    wave_sync();
    h_rank1(L, rdn);
    gg_rank1(L, 1.0);
    wave_sync();
    ACCT(C.sRead += 64ll * K + 64ll * MJ + 16);
    if (NSL > 2 && SL >= 2 && K > 64)
        ACCT(C.sRead += 8ll * ((long long)(K - 64) * 64 + (long long)(K - 64) * (K - 65) / 2));
    K += 1;
    return true;
}

// Border rows after the deletion of row p without a re-sweep (one register slot, K <= 63).  The rank-1 update gives
// L33' = L33 L~ with L~(r,k) = p_r beta_k below the diagonal, so the rows k > p of every border column follow from
//   v_k = y_k + p_k y_p,   y'_k = v_k - p_k s_k,   s_{k+1} = a_k s_k + beta_k v_k  (s_{p+1} = 0),  a_k = d_k / d'_k:
// a first-order recurrence = a scan of affine maps over the lanes (six ds_bpermute steps; the multipliers a are shared
// by all columns).  The results stay in the OLD row positions; the caller shifts the rows up.
__device__ __forceinline__ void border_update_scan(Rows &R, int K, int p, const double (&pv)[NSL], const double (&bt)[NSL],
                                                   const double (&dgold)[NSL], int MJ) {
    const int lane = lane_id();
    const bool on = lane > p && lane < K;
    // inclusive scan of the multipliers; Astep[st] = the multiplier a lane applies to what step st brings in
    double A = on ? dgold[0] * R.rd[0] : 1.0;
    double Astep[6];
    sfor<0, 6>(SFOR_BODY(st) {
        SFOR_IDX(st);
        Astep[st] = A;
        A = A * scan_src_f64<st>(A, 1.0);
    });
    const double pk = on ? pv[0] : 0.0, bk = on ? bt[0] : 0.0;
#pragma unroll
    for (int w = 0; w < NR; ++w) {
        if (w < MJ || w == CC) {  // uniform
            const double yp = rbcast<1>(R.Y[w], p);
            const double v = on ? fma(pk, yp, R.Y[w][0]) : 0.0;
            double B = bk * v;
            sfor<0, 6>(SFOR_BODY(st) {
                SFOR_IDX(st);
                B = fma(Astep[st], scan_src_f64<st>(B, 0.0), B);
            });
            const double sk = lane_prev_f64(B, 0.0);  // exclusive: the state before row k
            R.Y[w][0] = on ? fma(-pk, sk, v) : R.Y[w][0];
        }
    }
}

// The variable of row p is about to leave F for a NONZERO bound: c = hq[F] has changed by dz * V[F, j] (hq itself is up
// to date).  V[F, j] is column p of V_FF = L D L', so L^-1 V[F, j] = D L' e_p: the border column y_c and t = H[:, c]
// follow from row p of the factor (before the row is deleted):  y_c,r += dz d_r L(p,r) (r <= p),  H[w][c] += dz X[w]_p.
template <int SL>
__device__ __forceinline__ void fold_block_shift(const WLds &L, Rows &R, int K, int p, int MJ, double dz, double xp) {
    const int lane = lane_id();
    double lrow[NSL];
    load_rowT<SL>(L.F, p, lrow);
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + KSLOT * t;
        const double lr = (r == p) ? 1.0 : ((r < p) ? lrow[t] : 0.0);
        R.Y[CC][t] = (r <= p) ? fma(dz * R.dg[t], lr, R.Y[CC][t]) : R.Y[CC][t];
    }
    if (lane < MJ) {  // lane w: xp = [A;G][w, j]
        const double hv = fma(dz, xp, L.H[lane * NR + CC]);
        L.H[lane * NR + CC] = hv;
        L.H[CC * NR + lane] = hv;
    }
    wave_sync();
    (void)K;
}

// Delete row p (its variable left F).  single: the border rows and H are still those of the current factor, so
// H follows by the block-inverse downdate; otherwise the caller re-forms H.
template <int SL>
__device__ __forceinline__ void delete_var(const WLds &L, Rows &R, int &K, int p, int MJ, bool downdate, bool scan,
                                           double &gz, double xp) {
    const int lane = lane_id();
    double pv[NSL], bt[NSL], rdold[NSL], dgold[NSL];
#pragma unroll
    for (int t = 0; t < NSL; ++t) {
        rdold[t] = R.rd[t];
        dgold[t] = R.dg[t];
    }
    constexpr bool STREAM = NSL > 2 && SL >= 2;  // (big-factor build: update and compaction in one streamed pass)
    constexpr bool MERGE = !STREAM && SL == 1;   // (one row slot: the update writes the compacted positions itself)
    if (STREAM) {
        if (K <= 192) delete_stream<SL, 1>(L, R, K, p, pv, bt);
        else delete_stream<SL, 2>(L, R, K, p, pv, bt);
    } else {
        delete_update<SL, MERGE>(L.F, R, K, p, pv, bt);
    }
    if (downdate) {
        // H' = H - g g' / m,  g = [A;G c']' V_FF^-1 e_p = Y' D^-1 f,  m = (V_FF^-1)_pp = f' D^-1 f,
        // f = L^-1 e_p: f_p = 1, f_r = -p_r below (p = L33^-1 l, the vector the rank-1 update walks through)
        double fr[NSL], frd[NSL], msum = 0.0;
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int r = lane + KSLOT * t;
            fr[t] = (r == p) ? 1.0 : ((r > p && r < K) ? -pv[t] : 0.0);
            frd[t] = fr[t] * rdold[t];
            msum = fma(fr[t], frd[t], msum);
        }
        const double mpp = wave_sum(msum);
        double prod[NR], sums[NR];
#pragma unroll
        for (int w = 0; w < NR; ++w) {
            double s = 0.0;
            if (w < MJ || w == CC) {  // uniform
#pragma unroll
                for (int t = 0; t < SL; ++t) s = (lane + KSLOT * t < K) ? fma(R.Y[w][t], frd[t], s) : s;
            }
            prod[w] = s;
        }
        wave_sum_multi<NR>(prod, sums);
        {
            double gv = 0.0;
#pragma unroll
            for (int w = 0; w < NR; ++w) gv = (lane == w) ? sums[w] : gv;
            if (lane < NR) L.yn[lane] = gv;
        }
        wave_sync();
        h_rank1(L, -1.0 / mpp);
        wave_sync();
    }
    if (SL == 1 && scan) border_update_scan(R, K, p, pv, bt, dgold, MJ);  // (after the downdate: that one needs the old rows)
    {   // the Gram matrix of the rows of [A;G][:, F] loses the column of the deleted variable (xp: lane w = [A;G][w, j])
        if (lane < 16) L.xn[lane] = (lane < MJ) ? xp : 0.0;
        gz = fma(-xp, rbcast<SL>(R.zF, p), gz);  // [A;G][:, F] z_F loses the variable's term
        wave_sync();
        gg_rank1(L, -1.0);
        wave_sync();
    }
    if (!STREAM) delete_compact<SL, MERGE>(L.F, K, p);
    // per-row registers: rows above p move up
    const int rp = rbcast_i<SL>(R.rank, p);
    shift_up_i<SL>(R.ord, p);
    shift_up_i<SL>(R.rank, p);
    shift_up<SL>(R.zF, p);
    shift_up<SL>(R.ur, p);
    shift_up<SL>(R.dr, p);
    shift_up<SL>(R.dg, p);
    shift_up<SL>(R.rd, p);
    if (SL == 1 && scan) {
#pragma unroll
        for (int w = 0; w < NR; ++w)
            if (w < MJ || w == CC) shift_up<SL>(R.Y[w], p);
    }
    K -= 1;
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + KSLOT * t;
        R.rank[t] = (r < K && R.rank[t] > rp) ? R.rank[t] - 1 : R.rank[t];
    }
}

// hq = q + sum over the bound variables with z != 0 of V[:,i] z_i;  bEall = rhs - [A;G] zB   (SSQP.jl:295,324)
__device__ __forceinline__ void refresh_caches(WCtx &C, double2 (&hq)[NCH], double &bEv, const double *zg,
                                               unsigned Sp) {
    const int lane = lane_id();
    const int N = C.N, MJ = C.MJ;
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const int r = 2 * lane + 128 * m;
        hq[m] = (r < N) ? *reinterpret_cast<const double2 *>(C.q + r) : make_double2(0.0, 0.0);
    }
    // zB: z of the bound variables, 0 for the free ones
    double2 zb[NCH];
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const int r = 2 * lane + 128 * m;
        const double zx = __hip_atomic_load(zg + (r < N ? r : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double zy = __hip_atomic_load(zg + (r < N ? r + 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        zb[m].x = (r < N && st_of(Sp, 2 * m) != SSQP_IN) ? zx : 0.0;
        zb[m].y = (r < N && st_of(Sp, 2 * m + 1) != SSQP_IN) ? zy : 0.0;
    }
    [[maybe_unused]] int ncol = 0;
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        unsigned long long mx = __ballot(zb[m].x != 0.0), my = __ballot(zb[m].y != 0.0);
        unsigned long long any = mx | my;
        while (any) {  // by increasing index
            const int l = __ffsll((long long)any) - 1;
            any &= any - 1;
            if ((mx >> l) & 1ull) {
                axpy_dense(hq, C.V + (size_t)(2 * l + 128 * m) * N, readlane_f64(zb[m].x, l), N);
                ncol += 1;
            }
            if ((my >> l) & 1ull) {
                axpy_dense(hq, C.V + (size_t)(2 * l + 128 * m + 1) * N, readlane_f64(zb[m].y, l), N);
                ncol += 1;
            }
        }
    }
    double be = 0.0;
    for (int w = 0; w < MJ; ++w) {
        const double *__restrict__ row = C.Ct + (size_t)w * N;
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            const int r = 2 * lane + 128 * m;
            const double2 v = *reinterpret_cast<const double2 *>(row + (r < N ? r : 0));
            s = (r < N) ? fma(v.y, zb[m].y, fma(v.x, zb[m].x, s)) : s;
        }
        s = wave_sum(s);
        const double b = C.rhs[w] - s;
        be = (lane == w) ? b : be;
    }
    bEv = be;
    ACCT(C.sRead += 8ll * N * (ncol + MJ + 1));
}

// The per-QP state that lives across passes
struct WState {
    Rows R;
    double2 hq[NCH];
    double *zg;         // z of this QP in global memory (the output array): the live copy for the BOUND variables
                        // (x0 at the start, the bound a variable was snapped to since); free variables: R.zF
    unsigned Sp;        // statuses of this lane's 8 variables, 4 bits each
    unsigned Emask;     // active inequalities (bit j: S[N+j] == EO)
    double bEv;         // lane w: bEall_w = rhs_w - ([A;G] zB)_w
    double gz;          // lane w: ([A;G][:, F] z_F)_w, carried along with z_F (append, delete, blocked and full steps)
    int K;
    int nShift;         // status switches since hq / bEall were last re-evaluated from (z, S)
    bool hbValid, cDirty;
    // pending changes of F decided by the last pass
    unsigned long long del[NSL];    // rows to delete (lane mask per slot)
    int appJ;                       // variable to append, or -1
    double relDz;                   // the shift of z[appJ] that hq has already followed (0: none)
    double blkDz;                   // the same for the single variable a blocked step sent to a nonzero bound
    unsigned certMask;              // active-row set the full-rank certificate currently holds for (0: none); cleared
                                    // by a deletion from F, kept by appends, valid for every subset of its rows
    bool appAll;                    // append every variable with status IN that has no row (start, after freeK!)
};

// gamma += sum over the K free columns of V (weights alpha) and the W kept rows of [A;G] (weights alphaL), the columns
// streamed through the LDS ring (big-factor build).  NI = pieces of 1 KiB per column: N <= 128 NI.
template <int SL, int NI>
__device__ __forceinline__ void gamma_stream(const WCtx &C, const WLds &L, const Rows &R, int Kin, int Win,
                                             const double (&alpha)[NSL], double alRow, int raLane, double2 (&gam)[NCH]) {
    constexpr int D = RING_BYTES / (1024 * NI);
    static_assert(NI <= NCH, "a column has at most NCH pieces");
    const int lane = lane_id();
    const int N = uni(C.N), ncol = uni(Kin + Win), K = uni(Kin);
    auto colptr = [&](int e) -> const double * {
        if (e < K) return C.V + (size_t)rbcast_i<SL>(R.ord, e) * N;  // (uniform e)
        return C.Ct + (size_t)__builtin_amdgcn_readlane(raLane, e - K) * N;
    };
    auto weight = [&](int e) -> double {
        if (e < K) return rbcast<SL>(alpha, e);
        return readlane_f64(alRow, __builtin_amdgcn_readlane(raLane, e - K));
    };
    auto read = [&](int e, double2 (&v)[NI]) {
        const double *__restrict__ slot = L.ring + (size_t)(e % D) * (128 * NI);
#pragma unroll
        for (int m = 0; m < NI; ++m) v[m] = *reinterpret_cast<const double2 *>(slot + 128 * m + 2 * lane);
    };
    unsigned vo[NI];
    dense_offsets<NI>(N, vo);
    for (int e = 0; e < D && e < ncol; ++e) ring_issue_dense<NI>(L, e, colptr(e), vo);
    double2 vc[NI], vn[NI];
    wait_all_landed();
    read(0, vc);
    for (int e = 0; e < ncol; ++e) {
        if (e + 1 < ncol) {
            if (e + D <= ncol) wait_vm<NI * (D - 2)>();
            else wait_vm<0>();
            read(e + 1, vn);
        }
        const double wj = weight(e);
#pragma unroll
        for (int m = 0; m < NI; ++m) {  // (lanes beyond N accumulate junk: KKTchk! skips them)
            gam[m].x = fma(vc[m].x, wj, gam[m].x);
            gam[m].y = fma(vc[m].y, wj, gam[m].y);
        }
        wait_lds();
        if (e + D < ncol) ring_issue_dense<NI>(L, e % D, colptr(e + D), vo);
#pragma unroll
        for (int m = 0; m < NI; ++m) vc[m] = vn[m];
    }
}

// entry e (uniform, e < 128) of a list held two entries per lane (lane l: entries l and 64 + l): no LDS round trip
__device__ __forceinline__ int list_get(const int (&v)[3], int e) {
    const int a = __builtin_amdgcn_readlane(v[0], e & 63), b = __builtin_amdgcn_readlane(v[1], e & 63);
    const int c = __builtin_amdgcn_readlane(v[2], e & 63);
    return e < 64 ? a : (e < 128 ? b : c);
}

// The same gamma when the free variables outnumber the bound ones (K > N - K): V is symmetric, so V[b, F] alpha is the
// dot product of COLUMN b with alpha scattered to a dense N-vector -- one column per BOUND variable instead of one per
// free variable (the multipliers are only ever looked at on the bound variables, SSQP.jl:139-147).  The kept rows of
// [A;G] go in as before.  Four columns per round: their sums come out of one multi-vector butterfly and are added to the
// lanes that own those elements of gam.  N <= 128 NI.
// ROWS = false: the rows of [A;G] are in gam already; Sp then marks (as "not IN") exactly the variables whose column is wanted.
template <int SL, int NI, bool ROWS = true>
__device__ __forceinline__ void gamma_dot_stream(const WCtx &C, const WLds &L, const Rows &R, int Kin, int Win,
                                                 const double (&alpha)[NSL], double alRow, int raLane, double2 (&gam)[NCH],
                                                 unsigned Sp) {
    constexpr int D = RING_BYTES / (1024 * NI), DB = D / 4;
    const int lane = lane_id();
    const int N = uni(C.N), K = uni(Kin);
    // ---- alpha as a dense vector in registers, through the (idle) ring memory: zero, scatter by variable index, read back
    double *scr = const_cast<double *>(L.ring);
    double2 ad[NI];
#pragma unroll
    for (int m = 0; m < NI; ++m) *reinterpret_cast<double2 *>(scr + 128 * m + 2 * lane) = make_double2(0.0, 0.0);
    wave_sync();
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + KSLOT * t;
        if (r < K) scr[R.ord[t]] = alpha[t];
    }
    wave_sync();
#pragma unroll
    for (int m = 0; m < NI; ++m) ad[m] = *reinterpret_cast<const double2 *>(scr + 128 * m + 2 * lane);
    // ---- the bound variables (any order: the sums land on the elements' owners), listed in LDS (tr is idle here)
    int16_t *blist = reinterpret_cast<int16_t *>(L.tr);
    int nb = 0;
#pragma unroll
    for (int m = 0; m < NI; ++m) {
        const int i0 = 2 * lane + 128 * m;
        const bool bx = i0 < N && st_of(Sp, 2 * m) != SSQP_IN, by = i0 + 1 < N && st_of(Sp, 2 * m + 1) != SSQP_IN;
        const unsigned long long mx = __ballot(bx), my = __ballot(by);
        const unsigned long long lt = (1ull << lane) - 1ull;
        const int pos = nb + __popcll(mx & lt) + __popcll(my & lt);
        if (bx) blist[pos] = (int16_t)i0;
        if (by) blist[pos + (bx ? 1 : 0)] = (int16_t)(i0 + 1);
        nb += __popcll(mx) + __popcll(my);
    }
    nb = uni(nb);
    wave_sync();
    // ---- hq + [A;G][kept,:]' alphaL by the column stream (no free column: K = 0 there)
    if (ROWS) gamma_stream<SL, NI>(C, L, R, 0, Win, alpha, alRow, raLane, gam);
    if (nb == 0) return;
    // ---- the bound columns, four per round
    const int ngrp = (nb + 3) >> 2;
    int bv[3];  // (the list in registers, three entries per lane: no LDS round trip per column address)
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        const int e = lane + 64 * h;
        bv[h] = blist[e < nb ? e : nb - 1];
    }
    auto colof = [&](int e) -> const double * {  // (entries beyond the list repeat the last column; their sums are dropped)
        return C.V + (size_t)list_get(bv, e < nb ? e : nb - 1) * N;
    };
    unsigned vo[NI];
    dense_offsets<NI>(N, vo);
    for (int g = 0; g < DB && g < ngrp; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u) ring_issue_dense<NI>(L, (4 * g + u) % D, colof(4 * g + u), vo);
    wait_all_landed();
    for (int g = 0; g < ngrp; ++g) {
        if (g + DB <= ngrp) wait_vm<NI * (D - 4)>();  // (DB - 1 younger groups are in flight)
        else wait_vm<0>();
        double prod[4], sums[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double *__restrict__ slot = L.ring + (size_t)((4 * g + u) % D) * (128 * NI);
            double sacc = 0.0;
#pragma unroll
            for (int m = 0; m < NI; ++m) {
                const double2 v = *reinterpret_cast<const double2 *>(slot + 128 * m + 2 * lane);
                const bool in = 128 * m + 2 * lane < N;  // (N even: both halves; lanes beyond N hold stale LDS)
                sacc = in ? fma(v.y, ad[m].y, fma(v.x, ad[m].x, sacc)) : sacc;
            }
            prod[u] = sacc;
        }
        wait_lds();
        if (g + DB < ngrp) {
#pragma unroll
            for (int u = 0; u < 4; ++u) ring_issue_dense<NI>(L, (4 * g + u) % D, colof(4 * (g + DB) + u), vo);
        }
        wave_sum_multi<4>(prod, sums);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (4 * g + u < nb) {  // uniform
                const int b = list_get(bv, 4 * g + u);
                const int k = dk_of(b);
                const bool me = lane == dl_of(b);
#pragma unroll
                for (int m = 0; m < NI; ++m) {
                    gam[m].x += (me && k == 2 * m) ? sums[u] : 0.0;
                    gam[m].y += (me && k == 2 * m + 1) ? sums[u] : 0.0;
                }
            }
        }
    }
}

// ---- fp32 screening of the multiplier pass (big-factor build, N <= 256) ----
// KKTchk! (SSQP.jl:136-188) releases ONE bound variable, the one with the smallest event value L (L = -gamma at an upper
// bound, gamma at a lower one) among those below -tolG; the pass streams a column of V per bound variable for it -- the
// largest single stream of a big-factor pass.  Here the columns come from an fp32 COPY of V (half the bytes, packed fp32
// arithmetic) and only decide WHICH columns are worth an exact look: with |g32_b - gamma_b| <= band for every b (band from
// the rounding analysis below, twice over), the variable KKTchk! picks is among those with L32 <= min L32 + 2 band, and
// no variable with L32 >= -tolG + band is a violator at all.  Those few columns (one to three, typically) are then
// formed exactly, in f64, as before: every decision is taken on exact values, the fp32 pass only prunes.
// band: g32 = fl(fl(gam0) + sum_f fl(v_bf) fl(alpha_f)) in fp32, any order; with T = vmax * sum|alpha| >= sum|v alpha|:
// conversions 2 u T, products u T, K - 1 adds (K - 1) u T, gam0's conversion and final add 2 u |gam0| + u T, u = 2^-24
// => |g32 - gamma| <= u ((K + 4) T + 2 max|gam0|); band is twice that.
// Returns the candidates as a mask over the lane's 8 variables; ncand = their number (-1: too many, take the exact pass).
template <int SL>
__device__ __forceinline__ unsigned gamma32_screen(WCtx &C, const WLds &L, const Rows &R, int Kin, const double (&alpha)[NSL],
                                                   const double2 (&gam)[NCH], unsigned Sp, double tolG, int &ncand) {
    const int lane = lane_id();
    const int N = uni(C.N), K = uni(Kin);
    float *scr = reinterpret_cast<float *>(const_cast<double *>(L.ring) + RING_BYTES / 8);   // behind the ring: alpha32 (1 KiB), g32 (1 KiB)
    float *g32L = scr + 256;
    // ---- alpha as a dense fp32 vector (piece layout: lane l holds elements 4 l .. 4 l + 3), sum |alpha|
    *reinterpret_cast<float4 *>(scr + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4 *>(g32L + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
    wave_sync();
    double a1 = 0.0;
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + KSLOT * t;
        if (r < K) {
            scr[R.ord[t]] = (float)alpha[t];
            a1 += fabs(alpha[t]);
        }
    }
    a1 = wave_sum(a1);
    wave_sync();
    const float4 ad = *reinterpret_cast<const float4 *>(scr + 4 * lane);
    // ---- the bound variables, listed in LDS (tr is idle here), then in registers
    int16_t *blist = reinterpret_cast<int16_t *>(L.tr);
    int nb = 0;
#pragma unroll
    for (int m = 0; m < 2; ++m) {   // (N <= 256: two chunks)
        const int i0 = 2 * lane + 128 * m;
        const bool bx = i0 < N && st_of(Sp, 2 * m) != SSQP_IN, by = i0 + 1 < N && st_of(Sp, 2 * m + 1) != SSQP_IN;
        const unsigned long long mx = __ballot(bx), my = __ballot(by);
        const unsigned long long lt = (1ull << lane) - 1ull;
        const int pos = nb + __popcll(mx & lt) + __popcll(my & lt);
        if (bx) blist[pos] = (int16_t)i0;
        if (by) blist[pos + (bx ? 1 : 0)] = (int16_t)(i0 + 1);
        nb += __popcll(mx) + __popcll(my);
    }
    nb = uni(nb);
    wave_sync();
    if (nb > 0) {
        const int ngrp = (nb + 3) >> 2;
        int bv[3];
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            const int e = lane + 64 * h;
            bv[h] = blist[e < nb ? e : nb - 1];
        }
        // The columns come straight into REGISTERS (one 16-byte load per lane and column), three groups of four in flight in
        // three register sets that change roles: an LDS-DMA piece costs 100+ cycles to issue and its data another LDS read,
        // which is what bounded this loop at ~400 cycles per column whatever the column's size
        const bool in = 4 * lane < N;   // (N a multiple of 4)
        const float *vbase = C.V32 + (in ? 4 * lane : 0);
        auto load4 = [&](int g, float4 (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = 4 * g + u;
                v[u] = *reinterpret_cast<const float4 *>(vbase + (size_t)list_get(bv, e < nb ? e : nb - 1) * N);
            }
        };
        auto use4 = [&](int g, const float4 (&v)[4]) __attribute__((always_inline)) {
            float prod[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float sacc = fmaf(v[u].w, ad.w, fmaf(v[u].z, ad.z, fmaf(v[u].y, ad.y, v[u].x * ad.x)));
                prod[u] = in ? sacc : 0.f;
            }
            // four wavefront sums without an LDS round trip in the chain: two halving steps (lane pairs, then quads: a lane
            // keeps ONE column's partial sum, class (b0, b1) = column 2 b0 + b1), the in-row steps, then the four rows are
            // added up through v_readlane; lane 0 stores the totals by variable id
            const bool b0 = lane & 1, b1 = lane & 2;
            auto fdpp = [](float x, auto ctrl) __attribute__((always_inline)) {
                return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, false));
            };
            const float r0 = (b0 ? prod[2] : prod[0]) + fdpp(b0 ? prod[0] : prod[2], IC<DPP_XOR1>{});
            const float r1 = (b0 ? prod[3] : prod[1]) + fdpp(b0 ? prod[1] : prod[3], IC<DPP_XOR1>{});
            float q = (b1 ? r1 : r0) + fdpp(b1 ? r0 : r1, IC<DPP_XOR2>{});
            q += fdpp(q, IC<0x124>{});  // row_ror:4
            q += fdpp(q, IC<0x128>{});  // row_ror:8   (every lane of a class now holds its row's sum)
            float tot[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cl = ((u >> 1) & 1) | ((u & 1) << 1);   // the lane (within a quad) of class (b0, b1) = (u >> 1, u & 1)
                float t = 0.f;
#pragma unroll
                for (int row = 0; row < 4; ++row) t += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q), 16 * row + cl));
                tot[u] = t;
            }
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (4 * g + u < nb) g32L[list_get(bv, 4 * g + u) & 255] = tot[u];
            }
        };
        float4 va[4], vb[4], vc[4];
        load4(0, va);
        load4(1, vb);
        load4(2, vc);
        for (int g = 0; g < ngrp; g += 3) {  // (groups beyond the list repeat its last column: their sums are dropped)
            use4(g, va);
            load4(g + 3, va);
            if (g + 1 < ngrp) use4(g + 1, vb);
            load4(g + 4, vb);
            if (g + 2 < ngrp) use4(g + 2, vc);
            load4(g + 5, vc);
        }
    }
    wave_sync();
    // ---- screening on the lane's eight variables (dense layout: elements 2 l, 2 l + 1 of chunks 0, 1)
    double gmx = 0.0;
    float lv[4];
    bool bd[4];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int i0 = 2 * lane + 128 * m;
        const float2 gs = *reinterpret_cast<const float2 *>(g32L + (i0 < N ? i0 : 0));
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int s = st_of(Sp, 2 * m + e);
            const double g0 = e ? gam[m].y : gam[m].x;
            const float g = (float)g0 + (e ? gs.y : gs.x);
            bd[2 * m + e] = i0 + e < N && s != SSQP_IN;
            lv[2 * m + e] = (s == SSQP_UP) ? -g : g;
            gmx = bd[2 * m + e] ? fmax(gmx, fabs(g0)) : gmx;
        }
    }
    gmx = wave_max(gmx);
    const double band = 0x1.0p-23 * ((double)(K + 4) * C.vmax * a1 + 2.0 * gmx);
    const double thr = -tolG + band;
    double lmin = INF;
#pragma unroll
    for (int k = 0; k < 4; ++k) lmin = (bd[k] && (double)lv[k] < thr) ? fmin(lmin, (double)lv[k]) : lmin;
    lmin = wave_min(lmin);
    unsigned cm = 0;
    int nc = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool c = bd[k] && (double)lv[k] < thr && (double)lv[k] <= lmin + 2.0 * band;
        cm |= c ? 1u << k : 0u;
        nc += __popcll(__ballot(c));
    }
    ACCT(C.sRead += 4ll * N * nb);
    C.nScreen += 1;
    ncand = nc > 48 ? -1 : nc;
    return cm;
}

// One pass for K > 0 with the kept factor in sync.  SSQP.jl:287-375.
template <int SL>
__device__ __forceinline__ int wave_pass(WCtx &C, const WLds &L, WState &S, double *__restrict__ gscr) {
    const int lane = lane_id();
    [[maybe_unused]] const int J = C.J;
    const int N = C.N, M = C.M, MJ = C.MJ;
    const double tol = C.tol, tolG = C.tolG;
    Rows &R = S.R;
    const int K = S.K;
    ssqp_trace *trace = (C.trace && C.iter <= C.ntrace) ? C.trace + (C.iter - 1) : nullptr;

    WPH_DECL;
    // ---- active rows (SSQP.jl:288-294): all equalities, then the inequalities with status EO
    const unsigned act = ((1u << M) - 1u) | (S.Emask << M);
    const int W0 = __popc(act);
    // ---- rank filter on [AE bE]  (SSQP.jl:310-319)
    unsigned kept = act;
    if (lane < MJX + 1) L.aLrow[lane] = 0.0;
    {   // active row ids in order (the kept ones, unless the filter below purges some)
        const unsigned below = act & ((1u << (lane & 31)) - 1u);
        if (lane < MJX && ((act >> lane) & 1u)) L.ra[__popc(below)] = (int16_t)lane;
    }
    wave_sync();
    // register copies of the two small LDS tables the pass keeps asking for (one read each instead of one per use):
    // raLane = L.ra[lane] (row id of kept position `lane`), alRow = L.aLrow[lane] (alphaL of row id `lane`, set below)
    int raLane = L.ra[lane < MJX ? lane : 0];
    if (W0 > 0) {
        // usually the Gram matrix of the active rows proves that the filter cannot purge anything
        bool certified = (act & ~S.certMask) == 0u;  // (a subset of a certified row set with no column lost since)
        if (!certified && K >= W0) {
            if (W0 > 8) certified = full_rank_certified<MJX>(L, W0, raLane);
            else if (W0 > 4) certified = full_rank_certified<8>(L, W0, raLane);
            else certified = full_rank_certified<4>(L, W0, raLane);
            S.certMask = certified ? act : 0u;
        }
        if (!certified) {
            if (SL == 1) kept = rank_filter<1>(R, S.bEv, act, K, tol, C.Ct, N);
            else kept = rank_filter<2>(R, S.bEv, act, K, tol, C.Ct, N);
            if (kept != act) {
                const unsigned below = kept & ((1u << (lane & 31)) - 1u);
                if (lane < MJX && ((kept >> lane) & 1u)) L.ra[__popc(below)] = (int16_t)lane;
                wave_sync();
                raLane = L.ra[lane < MJX ? lane : 0];
            }
        }
    }
    const int W = __popc(kept);
    WPH(0);  // rank filter (or the certificate that replaces it)
    // ---- Schur system (AE V^-1 AE') lam = bE + AE V^-1 c ; alphaL = -lam  (SSQP.jl:325-328, 351)
    if (W > 0) {
        double lam = 0.0;
        bool okH = true;
        if (W > 8) okH = schur_solve<MJX>(L, S.bEv, W, lam, raLane);
        else if (W > 6) okH = schur_solve<8>(L, S.bEv, W, lam, raLane);
        else if (W > 4) okH = schur_solve<6>(L, S.bEv, W, lam, raLane);
        else okH = schur_solve<4>(L, S.bEv, W, lam, raLane);
        if (!okH) {  // cholesky(C) of the reference throws (SSQP.jl:328)
            C.ret = -1;
            C.det = SSQP_DETAIL_POSDEF_C;
            return W_BREAK;
        }
        if (lane < W) L.aLrow[raLane] = -lam;
        wave_sync();
    }
    const double alRow = L.aLrow[lane < MJX ? lane : 0];
    WPH(1);  // Schur gather + lambda
    // ---- alpha = -V_FF^-1 (AE' alphaL + c):  v = D^-1 (Y_A alphaL + y_c), alpha = -L'^-1 v   (SSQP.jl:329-331)
    double v[NSL];
#pragma unroll
    for (int t = 0; t < NSL; ++t) v[t] = 0.0;
#pragma unroll
    for (int t = 0; t < SL; ++t) v[t] = R.Y[CC][t];
#pragma unroll
    for (int w = 0; w < MJX; ++w) {
        if ((kept >> w) & 1u) {  // uniform
            const double al = readlane_f64(alRow, w);
#pragma unroll
            for (int t = 0; t < SL; ++t) v[t] = fma(R.Y[w][t], al, v[t]);
        }
    }
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + KSLOT * t;
        v[t] = (r < K) ? v[t] * R.rd[t] : 0.0;
    }
    if (NSL > 2 && SL >= 2) {  // (rows >= 64 in global memory: by columns, streamed through the ring)
        if (K <= 128) back_sweep_rows<SL, 1>(L, K, v);
        else back_sweep_rows<SL, 2>(L, K, v);
        // (the factor's global part, rows max(64, c + 1) .. K - 1 of every column: read once here, once by the append)
        if (K > 64) ACCT(C.sRead += 8ll * ((long long)(K - 64) * 64 + (long long)(K - 64) * (K - 65) / 2));
    } else {
        back_sweep<SL>(L.F, K, v);
    }
    WPH(2);  // v + back substitution
    double alpha[NSL], p[NSL];
    double pa = 0.0;
    int pnan = 0;
#pragma unroll
    for (int t = 0; t < NSL; ++t) alpha[t] = p[t] = 0.0;
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + KSLOT * t;
        alpha[t] = -v[t];
        p[t] = (r < K) ? alpha[t] - R.zF[t] : 0.0;  // SSQP.jl:332
        if (p[t] != p[t]) pnan = 1;
        pa = fmax(pa, fabs(p[t]));
    }
    const double pinf = wave_max(pa);
    const bool anyNan = __ballot(pnan) != 0ull;

    {   // per-pass accounting
        [[maybe_unused]] const long long k = K, r = N - K, w = W;
        ACCT(C.sBytes += 8ll * (k * k + r * k) + 8ll * MJ * N + 48ll * N + 4ll * (N + J));
        ACCT(C.sFlops += k * k * k + 4 * k * k * w + 2 * k * k + 2 * r * k + 2ll * W0 * r + w * w * w);
        ACCT(C.sK3 += k * k * k);
    }

    WPH(3);  // p, norm, accounting
    if (pinf > tolG && !anyNan) {  // ------------------------ aStep!  SSQP.jl:61-134
        double Lr[NSL];
        double lmin = INF;
#pragma unroll
        for (int t = 0; t < NSL; ++t) Lr[t] = INF;
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int r = lane + KSLOT * t;
            if (r < K) {
                const double tt = p[t], h = R.zF[t];
                if (tt > tol && R.ur[t] < INF) Lr[t] = (R.ur[t] - h) / tt;         // :69-71
                else if (tt < -tol && R.dr[t] > -INF) Lr[t] = (R.dr[t] - h) / tt;  // :73-75
                lmin = fmin(lmin, Lr[t]);
            }
        }
        // inactive inequalities: zo = g - G z, po = G[:,F] p  (:78-89), with NO sum over the free variables: row w of
        // [A;G] lives in lane w; [A;G][w,F] alpha = -(sum over the kept rows a of H[w][a] alphaL_a + H[w][c]) comes
        // straight from the kept Schur block (alpha = -V_FF^-1 (AE' alphaL + c)), [A;G][w,F] z_F is carried along
        // (gz), and the bound part of G z is cached in bEall.
        double Ga = 0.0;
        {
            const int wl = lane < MJ ? lane : 0;
            double hw[MJX];  // (row wl of H read at once: a read under the uniform branch waits for its own round trip)
#pragma unroll
            for (int a = 0; a < MJX; ++a) hw[a] = L.H[wl * NR + a];
            Ga = L.H[wl * NR + CC];
#pragma unroll
            for (int a = 0; a < MJX; ++a)
                if ((kept >> a) & 1u) Ga = fma(hw[a], readlane_f64(alRow, a), Ga);  // uniform
            Ga = -Ga;
        }
        const double po = Ga - S.gz, zo = S.bEv - S.gz;
        const bool inact = lane >= M && lane < MJ && !((S.Emask >> ((lane - M) & 31)) & 1u);
        const double lin = (inact && po > tol) ? zo / po : INF;  // :85-86  (lane M + j: inequality j)
        lmin = fmin(lmin, lin);
        const double L1 = wave_min(lmin);
        ACCT(C.sFlops += 2ll * (J - __popc(S.Emask)) * (N + K));
        WPH(4);  // aStep ratios + min
        if (L1 < 1.0) {  // blocked  (:98-127)
            int firstId = 0x7fffffff;
            unsigned long long dm[NSL];
            bool hit[NSL];
            double zstep[NSL];  // z_F + L1 p before the snap
#pragma unroll
            for (int t = 0; t < NSL; ++t) {
                dm[t] = 0ull;
                hit[t] = false;
                zstep[t] = 0.0;
            }
#pragma unroll
            for (int t = 0; t < SL; ++t) {
                const int r = lane + KSLOT * t;
                if (r < K) {
                    double zn = R.zF[t] + L1 * p[t];  // :99
                    zstep[t] = zn;
                    if (Lr[t] < INF && !(Lr[t] - L1 > tol)) {  // :102-116
                        const bool up = p[t] > tol;
                        zn = up ? R.ur[t] : R.dr[t];
                        hit[t] = true;
                        firstId = min(firstId, R.ord[t] + 1);
                    }
                    R.zF[t] = zn;
                }
                dm[t] = __ballot(hit[t]);
            }
            const bool ihit = (lin < INF) && !(lin - L1 > tol);
            const unsigned long long im = __ballot(ihit);
            if (ihit) firstId = min(firstId, N + (lane - M) + 1);
            S.Emask |= (unsigned)(im >> M);
            S.gz = (lane < MJ) ? fma(L1, po, S.gz) : S.gz;  // z_F += L1 p
            // the switched variables: statuses, z of the bound set, and the caches hq / bEall follow
#pragma unroll
            for (int t = 0; t < SL; ++t) {
                unsigned long long mm = dm[t];
                while (mm) {  // by increasing row (any order gives the same set; hq is updated column by column)
                    const int l = __ffsll((long long)mm) - 1;
                    mm &= mm - 1;
                    const int jv = __builtin_amdgcn_readlane(R.ord[t], l);
                    const double pj = readlane_f64(p[t], l);
                    const double zn = readlane_f64(R.zF[t], l);
                    {   // the variable was snapped to its bound: gz follows the exact value of z_F (a rounding-sized
                        // difference for the variable that set the step length is not worth a memory access)
                        const double dzs = zn - readlane_f64(zstep[t], l);
                        if (fabs(dzs) > 0x1.0p-44 * fmax(1.0, fabs(zn))) {  // uniform
                            const double cjv = C.Ct[(size_t)(lane < MJ ? lane : 0) * N + jv];
                            S.gz = (lane < MJ) ? fma(cjv, dzs, S.gz) : S.gz;
                        }
                    }
                    st_set(S.Sp, jv, (pj > tol) ? SSQP_UP : SSQP_DN);
                    if (lane == 0) S.zg[jv] = zn;
                    if (zn != 0.0) {
                        bound_shift(C, S.hq, S.bEv, jv, zn);
                        S.nShift += 1;
                        S.blkDz = zn;  // (used only when this is the pass's single deletion)
                        S.cDirty = true;
                        ACCT(C.sRead += 8ll * N + 64ll * MJ);
                    }
                }
            }
            WPH(5);  // blocked: switches + bound shifts
            int nblk = 0;
#pragma unroll
            for (int t = 0; t < NSL; ++t) {
                S.del[t] = dm[t];
                nblk += __popcll(dm[t]);
            }
            if (nblk != 1) S.blkDz = 0.0;
            if (trace) {
                int f = firstId;
                f = min(f, dpp_i32<DPP_XOR1>(f));
                f = min(f, dpp_i32<DPP_XOR2>(f));
                f = min(f, dpp_i32<DPP_HALF_MIRROR>(f));
                f = min(f, dpp_i32<DPP_MIRROR>(f));
                f = min(f, __builtin_amdgcn_update_dpp(f, f, 0x142, 0xA, 0xF, false));
                f = min(f, __builtin_amdgcn_update_dpp(f, f, 0x143, 0xC, 0xF, false));
                f = __builtin_amdgcn_readlane(f, 63);
                if (lane == 0) *trace = ssqp_trace{K, W, 1, f};
            }
            return W_CONTINUE;
        }
        // full step: z[F] = alpha  (:130)
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int r = lane + KSLOT * t;
            R.zF[t] = (r < K) ? alpha[t] : R.zF[t];
        }
        S.gz = (lane < MJ) ? Ga : S.gz;
    }

    WPH(6);  // full step bookkeeping
    // ---- multipliers: gamma = V[B,F] alpha + V[B,B] zB + q[B] + AB' alphaL  (SSQP.jl:352)
    //      = hq + V[:,F] alpha + [A;G][kept,:]' alphaL on the bound variables
    double2 gam[NCH];
#pragma unroll
    for (int m = 0; m < NCH; ++m) gam[m] = S.hq[m];
    unsigned candMask = 0xFFu;   // the lane's variables KKTchk! looks at (all, unless the fp32 screening pruned)
    bool screened = false;
    if (NSL > 2 && SL >= 2 && C.V32 != nullptr) {
        // rows of [A;G] first (exact), then the fp32 screening of the bound columns, then the few exact columns it asks for
        gamma_stream<SL, 2>(C, L, R, 0, W, alpha, alRow, raLane, gam);
        WPH(7);   // (rows of [A;G])
        int ncand = 0;
        const unsigned cm = gamma32_screen<SL>(C, L, R, K, alpha, gam, S.Sp, tolG, ncand);
        WPH(15);  // fp32 screening of the bound columns
        if (ncand >= 0) {
            screened = true;
            candMask = cm;
            if (ncand > 0) {  // exact gamma of the candidates: their columns only (a status word that marks just them as bound)
                unsigned spc = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) spc |= (unsigned)(((cm >> k) & 1u) ? SSQP_DN : SSQP_IN) << (4 * k);
                gamma_dot_stream<SL, 2, false>(C, L, R, K, W, alpha, alRow, raLane, gam, spc);
            }
            C.nCand += ncand;
            ACCT(C.sRead += 8ll * N * (ncand + W));
        } else {
            gamma_dot_stream<SL, 2, false>(C, L, R, K, W, alpha, alRow, raLane, gam, S.Sp);   // (every bound column, exactly)
            ACCT(C.sRead += 8ll * N * (N - K + W));
        }
    } else if (NSL > 2) {  // (big-factor build: the list below streamed through the LDS ring)
        if (SL >= 2 && N <= 256 && 2 * K > N) {  // fewer bound than free variables: by the bound columns (V is symmetric)
            if (N <= 128) gamma_dot_stream<SL, 1>(C, L, R, K, W, alpha, alRow, raLane, gam, S.Sp);
            else gamma_dot_stream<SL, 2>(C, L, R, K, W, alpha, alRow, raLane, gam, S.Sp);
            ACCT(C.sRead += 8ll * N * (N - K + W));
        } else {
            if (N <= 128) gamma_stream<SL, 1>(C, L, R, K, W, alpha, alRow, raLane, gam);
            else if (N <= 256) gamma_stream<SL, 2>(C, L, R, K, W, alpha, alRow, raLane, gam);
            else gamma_stream<SL, 4>(C, L, R, K, W, alpha, alRow, raLane, gam);
            ACCT(C.sRead += 8ll * N * (K + W));
        }
    } else {
        // one list: the K free columns of V (weights alpha) and the W kept rows of [A;G] (weights alphaL), NB at a
        // time with all their loads in flight together
        constexpr int NB = 4;
        const int ncolG = K + W;
        for (int e0 = 0; e0 < ncolG; e0 += NB) {
            const double *__restrict__ col[NB];
            double wj[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                const int e = (e0 + c < ncolG) ? e0 + c : e0;
                if (e < K) {  // uniform
                    col[c] = C.V + (size_t)rbcast_i<SL>(R.ord, e) * N;
                    wj[c] = rbcast<SL>(alpha, e);
                } else {
                    const int rid = __builtin_amdgcn_readlane(raLane, e - K);
                    col[c] = C.Ct + (size_t)rid * N;
                    wj[c] = readlane_f64(alRow, rid);
                }
                wj[c] = (e0 + c < ncolG) ? wj[c] : 0.0;
            }
            double2 vv[NCH][NB];
#pragma unroll
            for (int m = 0; m < NCH; ++m) {
                const int rr = 2 * lane + 128 * m;
#pragma unroll
                for (int c = 0; c < NB; ++c) vv[m][c] = *reinterpret_cast<const double2 *>(col[c] + (rr < N ? rr : 0));
            }
#pragma unroll
            for (int m = 0; m < NCH; ++m) {  // (lanes beyond N accumulate finite junk from element 0: KKTchk! skips them)
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    gam[m].x = fma(vv[m][c].x, wj[c], gam[m].x);
                    gam[m].y = fma(vv[m][c].y, wj[c], gam[m].y);
                }
            }
        }
        ACCT(C.sRead += 8ll * N * (K + W));
    }
    {
        [[maybe_unused]] const long long r = N - K;
        ACCT(C.sBytes += 8ll * r * r);
        ACCT(C.sFlops += 2ll * r * r + 2ll * r * K);
    }

    WPH(7);  // gamma pass
    // ---- KKTchk!  SSQP.jl:136-188
    KeyMin ev{INF, 0x7fffffff};
    double lamRow = alRow;  // lane w: the multiplier of row w of [A;G] as KKTchk! sees it (0: inactive / no value)
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 2 * lane + 128 * m + e;
            const double g = e ? gam[m].y : gam[m].x;
            const int s = st_of(S.Sp, 2 * m + e);
            if (i < N && ((candMask >> (2 * m + e)) & 1u)) {
                if (s == SSQP_UP && g > tolG) ev = keymin(ev, KeyMin{-g, i});        // :141-143
                else if (s == SSQP_DN && g < -tolG) ev = keymin(ev, KeyMin{g, i});  // :144-146
            }
        }
    }
    if (S.Emask != 0u) {  // multipliers of the active inequalities (:149-171)
        if (kept == act) {  // every active row kept its own multiplier
            if (lane >= M && lane < MJ && ((act >> lane) & 1u)) {
                const double Lda = alRow;
                if (Lda < -tolG) ev = keymin(ev, KeyMin{Lda, N + lane - M});
            }
        } else {
            // purged rows: Lda = alphaL' * (AE' \ GE[j,F])  (:158-159): least squares by modified Gram-Schmidt (two
            // passes) on the K x W matrix AE' in the wavefront's global scratch
            double *Q = gscr, *Rm = Q + (size_t)K * W, *yv = Rm + W * W;
            {
                int wi = 0;
#pragma unroll
                for (int w = 0; w < MJX; ++w) {
                    if ((kept >> w) & 1u) {
                        double xw[NSL];
                        gather_X<SL>(C.Ct, N, w, R.ord, K, xw);
#pragma unroll
                        for (int t = 0; t < SL; ++t) {
                            const int r = lane + KSLOT * t;
                            if (r < K) Q[r + (size_t)K * wi] = xw[t];
                        }
                        wi += 1;
                    }
                }
            }
            for (int e = lane; e < W * W; e += 64) Rm[e] = 0.0;
            wave_sync();
            for (int c = 0; c < W; ++c) {
                for (int pass = 0; pass < 2; ++pass)
                    for (int b2 = 0; b2 < c; ++b2) {
                        double s = 0.0;
                        for (int k = lane; k < K; k += 64) s = fma(Q[k + (size_t)K * b2], Q[k + (size_t)K * c], s);
                        s = wave_sum(s);
                        for (int k = lane; k < K; k += 64) Q[k + (size_t)K * c] = fma(-s, Q[k + (size_t)K * b2], Q[k + (size_t)K * c]);
                        if (lane == 0) Rm[b2 + W * c] += s;
                        wave_sync();
                    }
                double s = 0.0;
                for (int k = lane; k < K; k += 64) s = fma(Q[k + (size_t)K * c], Q[k + (size_t)K * c], s);
                s = sqrt(wave_sum(s));
                if (lane == 0) Rm[c + W * c] = s;
                const double rs = (s > 0.0) ? 1.0 / s : 0.0;
                for (int k = lane; k < K; k += 64) Q[k + (size_t)K * c] *= rs;
                wave_sync();
            }
#pragma unroll
            for (int w = 0; w < MJX; ++w) {
                if (w >= M && ((act >> w) & 1u)) {  // active inequality w - M (uniform)
                    double Lda;
                    if ((kept >> w) & 1u) {
                        Lda = L.aLrow[w];
                    } else {
                        double xw[NSL];
                        gather_X<SL>(C.Ct, N, w, R.ord, K, xw);
                        for (int c = 0; c < W; ++c) {  // yq = Q' gv
                            double s = 0.0;
#pragma unroll
                            for (int t = 0; t < SL; ++t) {
                                const int r = lane + KSLOT * t;
                                if (r < K) s = fma(Q[r + (size_t)K * c], xw[t], s);
                            }
                            s = wave_sum(s);
                            if (lane == 0) yv[c] = s;
                        }
                        wave_sync();
                        double s = 0.0;
                        if (lane == 0) {  // x = R^-1 yq ; Lda = alphaL . x
                            for (int c = W - 1; c >= 0; --c) {
                                double sv = yv[c];
                                for (int b2 = c + 1; b2 < W; ++b2) sv -= Rm[c + W * b2] * yv[b2];
                                yv[c] = (Rm[c + W * c] > 0.0) ? sv / Rm[c + W * c] : 0.0;
                            }
                            for (int c = 0; c < W; ++c) s = fma(L.aLrow[L.ra[c]], yv[c], s);
                        }
                        Lda = readlane_f64(s, 0);
                        wave_sync();
                    }
                    if (lane == 0 && Lda < -tolG) ev = keymin(ev, KeyMin{Lda, N + w - M});
                    lamRow = (lane == w) ? Lda : lamRow;
                }
            }
        }
    }
    WPH(8);  // KKT scan
    ev = wave_keymin(ev);
    if (ev.v < INF) {  // release the single tightest one (:175-184)
        if (ev.ord < N) {
            const int jv = ev.ord;
            st_set(S.Sp, jv, SSQP_IN);
            const double zr = __hip_atomic_load(S.zg + jv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            S.relDz = 0.0;
            if (zr != 0.0) {  // B loses a column with a nonzero weight
                bound_shift(C, S.hq, S.bEv, jv, -zr);
                S.nShift += 1;
                S.cDirty = true;
                S.relDz = -zr;
                ACCT(C.sRead += 8ll * N + 64ll * MJ);
            }
            S.appJ = jv;
        } else {
            S.Emask &= ~(1u << (ev.ord - N));
        }
        WPH(9);  // release + bound shift
        if (trace && lane == 0) *trace = ssqp_trace{K, W, 2, ev.ord + 1};
        return W_CONTINUE;
    }
    // ---- optimal: the multipliers of this pass leave the kernel when asked for (alphaL SSQP.jl:351, gamma :352)
    if (C.lamOut && lane < MJ) C.lamOut[lane] = lamRow;
    if (NSL > 2 && SL >= 2 && screened && C.gamOut) {  // (the screening formed only the candidates exactly: all of them now)
#pragma unroll
        for (int m = 0; m < NCH; ++m) gam[m] = S.hq[m];
        gamma_dot_stream<SL, 2, true>(C, L, R, K, W, alpha, alRow, raLane, gam, S.Sp);
    }
    if (C.gamOut) {
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            const int r = 2 * lane + 128 * m;
            if (r < N) {
                double2 gg;
                gg.x = (st_of(S.Sp, 2 * m) != SSQP_IN) ? gam[m].x : 0.0;
                gg.y = (st_of(S.Sp, 2 * m + 1) != SSQP_IN) ? gam[m].y : 0.0;
                *reinterpret_cast<double2 *>(C.gamOut + r) = gg;
            }
        }
    }
    // ---- polishSz!  SSQP.jl:10-32
    {
        unsigned long long sn[NSL];
        bool snap[NSL], sup[NSL];
#pragma unroll
        for (int t = 0; t < NSL; ++t) {
            sn[t] = 0ull;
            snap[t] = sup[t] = false;
        }
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int r = lane + KSLOT * t;
            if (r < K) {
                const double zi = R.zF[t];
                if (fabs(zi - R.dr[t]) < tol) {
                    R.zF[t] = R.dr[t];
                    snap[t] = true;
                } else if (fabs(zi - R.ur[t]) < tol) {
                    R.zF[t] = R.ur[t];
                    snap[t] = true;
                    sup[t] = true;
                }
            }
            sn[t] = __ballot(snap[t]);
            const unsigned long long su = __ballot(sup[t]);
            unsigned long long mm = sn[t];
            while (mm) {
                const int l = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                const int jv = __builtin_amdgcn_readlane(R.ord[t], l);
                st_set(S.Sp, jv, ((su >> l) & 1ull) ? SSQP_UP : SSQP_DN);
            }
        }
        unsigned em = 0;
#pragma unroll
        for (int w = 0; w < MJX; ++w) {
            if (w >= M && w < MJ) {  // S[N+j] = |g_j - G[j,:] z| < tol ? EO : OE   (:28-30)
                double sz = 0.0, xw[NSL];
                gather_X<SL>(C.Ct, N, w, R.ord, K, xw);
#pragma unroll
                for (int t = 0; t < SL; ++t) {
                    const int r = lane + KSLOT * t;
                    sz = (r < K) ? fma(xw[t], R.zF[t], sz) : sz;
                }
                sz = wave_sum(sz);
                const double res = readlane_f64(S.bEv, w) - sz;
                if (fabs(res) < tol) em |= 1u << (w - M);
            }
        }
        S.Emask = em;
    }
    if (trace && lane == 0) *trace = ssqp_trace{K, W, 3, 0};
    C.ret = C.iter;  // SSQP.jl:374
    return W_BREAK;
}

// bring the kept factor in line with the decisions of the last pass
template <int SL>
__device__ __forceinline__ int wave_sync_factor(WCtx &C, const WLds &L, WState &S) {
    const int lane = lane_id();
    Rows &R = S.R;
    const int MJ = C.MJ;
    WPH_DECL;
    int ndel = 0;
#pragma unroll
    for (int t = 0; t < NSL; ++t) ndel += __popcll(S.del[t]);
    if (ndel > 0) {
        S.certMask = 0u;  // (a column of AE goes: the Gram pivots may shrink)
        const bool single = (ndel == 1);
        // A single deletion with one register slot: the border rows follow by a scan and H by the downdate, and when
        // the variable went to a nonzero bound the change of c is folded in beforehand -- no re-gather, no re-sweep.
        const bool fast = single && SL == 1 && (!S.cDirty || S.blkDz != 0.0);
        if (fast) {
            const int pl = 63 - __clzll(S.del[0]);
            const int jp = rbcast_i<SL>(R.ord, pl);
            const double xp = C.Ct[(size_t)(lane < MJ ? lane : 0) * C.N + jp];  // lane w: [A;G][w, jp] (used after the update)
            if (S.cDirty) fold_block_shift<SL>(L, R, S.K, pl, MJ, S.blkDz, xp);
            delete_var<SL>(L, R, S.K, pl, MJ, true, true, S.gz, xp);
#pragma unroll
            for (int t = 0; t < NSL; ++t) S.del[t] = 0ull;
            S.cDirty = false;
            S.blkDz = 0.0;
            WPH(10);  // deletes (update + downdate + compaction + shifts)
            if (S.K == 0) {
                for (int e = lane; e < NR * NR; e += 64) L.H[e] = 0.0;
                for (int e = lane; e < MJX * MJX; e += 64) L.GG[e] = 0.0;
                wave_sync();
            }
        } else {
        // highest row first: deleting row p leaves the rows below p in place
        sfor<0, SL>(SFOR_BODY(tr) {
            SFOR_IDX(tr);
            constexpr int t = SL - 1 - tr;
            unsigned long long dm = S.del[t];
            while (dm) {
                const int pl = 63 - __clzll(dm);
                dm &= ~(1ull << pl);
                const int jp = rbcast_i<SL>(R.ord, pl + KSLOT * t);
                const double xp = C.Ct[(size_t)(lane < MJ ? lane : 0) * C.N + jp];
                delete_var<SL>(L, R, S.K, pl + KSLOT * t, MJ, single, false, S.gz, xp);
            }
        });
        WPH(10);  // deletes (update + downdate + compaction + shifts)
#pragma unroll
        for (int t = 0; t < NSL; ++t) S.del[t] = 0ull;
        S.blkDz = 0.0;
        if (S.K > 0) {
            const unsigned cols = ((1u << MJ) - 1u) | (1u << CC);
            border_sweep<SL>(L.F, R, S.K, cols, C.Ct, C.N, S.hq, &L);
            if (!single) recompute_H_all<SL>(L, R, S.K, MJ);
            else if (S.cDirty) recompute_H_c<SL>(L, R, S.K, MJ);
            S.cDirty = false;
        } else {
            for (int e = lane; e < NR * NR; e += 64) L.H[e] = 0.0;
            for (int e = lane; e < MJX * MJX; e += 64) L.GG[e] = 0.0;
            wave_sync();
        }
        }
    }
    if (ndel > 0) WPH(11);  // border sweep + H column after deletes
    if (S.appJ >= 0 || S.appAll) {
        double dzFold = 0.0;
        if (S.cDirty && S.K > 0) {  // c changed for the rows already in the factor
            if (S.appJ >= 0 && S.relDz != 0.0) {
                dzFold = S.relDz;  // ... by the released variable's column only: folded into its append
            } else {
                border_sweep<SL>(L.F, R, S.K, 1u << CC, C.Ct, C.N, S.hq, &L);
                recompute_H_c<SL>(L, R, S.K, MJ);
            }
        }
        WPH(12);  // c refresh before an append
        S.cDirty = false;
        S.relDz = 0.0;
        if (S.appJ >= 0) {
            if (S.K + 1 > C.RC) return W_HANDOVER;
            if (!append_var<SL>(C, L, R, S.K, S.appJ, S.hq, S.zg, dzFold, S.gz)) {
                C.ret = -1;
                C.det = SSQP_DETAIL_POSDEF_V;
                return W_BREAK;
            }
            WPH(13);  // one append
            S.appJ = -1;
        } else {  // every IN variable, by increasing index (findall order, SSQP.jl:276)
            S.appAll = false;
#pragma unroll
            for (int m = 0; m < NCH; ++m) {
                const unsigned long long mx = __ballot(2 * lane + 128 * m < C.N && st_of(S.Sp, 2 * m) == SSQP_IN);
                const unsigned long long my = __ballot(2 * lane + 128 * m < C.N && st_of(S.Sp, 2 * m + 1) == SSQP_IN);
                unsigned long long any = mx | my;
                while (any) {
                    const int l = __ffsll((long long)any) - 1;
                    any &= any - 1;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        if (((e ? my : mx) >> l) & 1ull) {
                            const int jv = 2 * l + 128 * m + e;
                            // (variables that already have a row are not in this state: the factor is empty)
                            if (S.K + 1 > C.RC || S.K + 1 > 64 * SL - 1) return W_HANDOVER;
                            if (!append_var<SL>(C, L, R, S.K, jv, S.hq, S.zg, 0.0, S.gz)) {
                                C.ret = -1;
                                C.det = SSQP_DETAIL_POSDEF_V;
                                return W_BREAK;
                            }
                        }
                    }
                }
            }
        }
    } else if (S.cDirty && S.K > 0) {  // only hq changed (cannot happen without a change of F today; kept for safety)
        border_sweep<SL>(L.F, R, S.K, 1u << CC, C.Ct, C.N, S.hq, &L);
        recompute_H_c<SL>(L, R, S.K, MJ);
        S.cDirty = false;
    }
    return W_CONTINUE;
}

// The second row slot (rows 64..127) parked in global scratch: park[f * 64 + lane], f = field
[[maybe_unused]] constexpr int PARK_FIELDS = 6 + NR;
__device__ __forceinline__ void slot1_store(const Rows &R, double *park) {
    const int lane = lane_id();
    park[0 * 64 + lane] = __hiloint2double(R.rank[1], R.ord[1]);
    park[1 * 64 + lane] = R.zF[1];
    park[2 * 64 + lane] = R.ur[1];
    park[3 * 64 + lane] = R.dr[1];
    park[4 * 64 + lane] = R.dg[1];
    park[5 * 64 + lane] = R.rd[1];
#pragma unroll
    for (int w = 0; w < NR; ++w) park[(6 + w) * 64 + lane] = R.Y[w][1];
}
__device__ __forceinline__ void slot1_load(Rows &R, const double *park) {
    const int lane = lane_id();
    const double oi = park[0 * 64 + lane];
    R.ord[1] = __double2loint(oi);
    R.rank[1] = __double2hiint(oi);
    R.zF[1] = park[1 * 64 + lane];
    R.ur[1] = park[2 * 64 + lane];
    R.dr[1] = park[3 * 64 + lane];
    R.dg[1] = park[4 * 64 + lane];
    R.rd[1] = park[5 * 64 + lane];
#pragma unroll
    for (int w = 0; w < NR; ++w) R.Y[w][1] = park[(6 + w) * 64 + lane];
}
__device__ __forceinline__ void slot1_reset(Rows &R) {  // the values every QP starts with
    R.ord[1] = 0; R.rank[1] = 0;
    R.zF[1] = R.ur[1] = R.dr[1] = 0.0;
    R.dg[1] = 1.0; R.rd[1] = 1.0;
#pragma unroll
    for (int w = 0; w < NR; ++w) R.Y[w][1] = 0.0;
}

// PARK: the build that keeps the second row slot in global scratch between the passes that need it.
template <bool PARK>
__device__ __forceinline__ void wave_solve_one(const SolveParams &P, int prob, const WLds &L, double *gscr, double *park) {
    const int lane = lane_id();
    const int N = P.N, M = P.M, J = P.J, MJ = P.MJ;
    WCtx C;
    C.N = N; C.M = M; C.J = J; C.MJ = MJ;
    C.tol = P.tol; C.tolG = P.tolG;
    C.V = P.V + (size_t)prob * P.sV;
    C.Ct = P.Ct + (size_t)prob * P.sCt;
    C.rhs = P.rhs + (size_t)prob * P.sRhs;
    C.q = P.q + (size_t)prob * P.sq;
    C.dlo = P.d + (size_t)prob * P.sd;
    C.uhi = P.u + (size_t)prob * P.su;
    C.trace = (!SSQP_LEAN_BUILD && P.trace) ? P.trace + (size_t)prob * P.ntrace : nullptr;
    C.ntrace = P.ntrace;
    C.lamOut = P.lamOut ? P.lamOut + (size_t)prob * MJ : nullptr;
    C.gamOut = P.gamOut ? P.gamOut + (size_t)prob * N : nullptr;
    C.RC = P.waveRC;
    // hand-over from a build with a smaller factor (P.resume): continue from its (z, S) at its pass count
    C.iter = (CAN_RESUME && P.resume) ? P.fbIter[prob] : 0;
    C.ret = 0; C.det = SSQP_DETAIL_NONE;
    C.sBytes = 0; C.sRead = 0; C.sFlops = 0; C.sK3 = 0; C.maxK = 0;
#ifdef SSQP_PHASE_PROFILE
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        C.ph[k] = 0;
        C.pn[k] = 0;
    }
    const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
#endif
    int32_t *Sg = P.S + (size_t)prob * (N + J);
    const double tol = P.tol;

    WState S;
    S.zg = P.z + (size_t)prob * N;
#pragma unroll
    for (int t = 0; t < NSL; ++t) {
        S.R.ord[t] = 0; S.R.rank[t] = 0;
        S.R.zF[t] = S.R.ur[t] = S.R.dr[t] = 0.0;
        S.R.dg[t] = 1.0; S.R.rd[t] = 1.0;
#pragma unroll
        for (int w = 0; w < NR; ++w) S.R.Y[w][t] = 0.0;
    }
    S.Sp = 0;
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const int r = 2 * lane + 128 * m;
        if (r < N && !(CAN_RESUME && P.resume))  // (resumed: z is the live copy the other build left)
            *reinterpret_cast<double2 *>(S.zg + r) = *reinterpret_cast<const double2 *>(P.x0 + (size_t)prob * N + r);
        const int s0 = (r < N) ? Sg[r] : SSQP_DN, s1 = (r < N) ? Sg[r + 1] : SSQP_DN;
        S.Sp |= ((unsigned)s0 & 15u) << (8 * m);
        S.Sp |= ((unsigned)s1 & 15u) << (8 * m + 4);
        S.hq[m] = make_double2(0.0, 0.0);
    }
    {
        const int sj = (lane < J) ? Sg[N + lane] : SSQP_OE;
        S.Emask = (unsigned)__ballot(sj == SSQP_EO);
    }
    S.bEv = 0.0;
    S.gz = 0.0;
    S.K = 0;
    S.hbValid = false;
    S.cDirty = false;
    S.nShift = 0;
#pragma unroll
    for (int t = 0; t < NSL; ++t) S.del[t] = 0ull;
    S.appJ = -1;
    S.relDz = 0.0;
    S.blkDz = 0.0;
    S.certMask = 0u;
    S.appAll = true;
    if (PARK) slot1_store(S.R, park);
    if (NSL > 2) {  // the factor's global part starts all zero and stays zero outside the factor (ring_issue_faccol)
        for (int e = lane; e < 256 * 192; e += 64) L.F.L1[e] = 0.0;
        for (int e = lane; e < 192 * LR_STRIDE; e += 64) L.F.LR[e] = 0.0;
    }
    C.V32 = nullptr;
    C.vmax = 0.0;
    C.nScreen = 0; C.nCand = 0;
    if (NSL > 2 && N <= 256 && (N & 3) == 0 && N >= 8) {
        // V rounded to fp32 for the screening of the multiplier pass (gamma32_screen): one read of V per QP (about two
        // passes' worth of bytes against ~140 passes), four columns' loads in flight; max |V_ij| on the way
        float *v32 = reinterpret_cast<float *>(gscr + WAVE_LS_DOUBLES_BIG + 256 * 192 + 64 + 192 * LR_STRIDE + 64);
        double vm = 0.0;
        for (int c0 = 0; c0 < N; c0 += 4) {
            double2 vv[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int r = 2 * lane + 128 * m;
                    vv[u][m] = *reinterpret_cast<const double2 *>(C.V + (size_t)(c0 + u) * N + (r < N ? r : 0));
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int r = 2 * lane + 128 * m;
                    if (r < N) {
                        *reinterpret_cast<float2 *>(v32 + (size_t)(c0 + u) * N + r) = make_float2((float)vv[u][m].x, (float)vv[u][m].y);
                        vm = fmax(vm, fmax(fabs(vv[u][m].x), fabs(vv[u][m].y)));
                    }
                }
        }
        C.vmax = wave_max(vm);
        C.V32 = v32;
    }
    for (int e = lane; e < NR * NR; e += 64) L.H[e] = 0.0;
    for (int e = lane; e < MJX * MJX; e += 64) L.GG[e] = 0.0;
    wave_sync();
    ACCT(C.sRead += 8ll * N + 4ll * (N + J));

    bool handover = false;
    for (;;) {
        C.iter += 1;
        if (C.iter > P.maxIter) {  // SSQP.jl:271-274
            C.ret = -C.iter;
            break;
        }
        ssqp_trace *trace = (C.trace && C.iter <= C.ntrace) ? C.trace + (C.iter - 1) : nullptr;
        // hq and bEall follow the status switches by one column each; rounding of those updates must not pile up
        // over a long run (the reference re-evaluates VBF'zB and bE in every pass): re-evaluate every 64 switches
        if (S.nShift >= 64) S.hbValid = false;
        if (!S.hbValid) {
            S.nShift = 0;
            S.relDz = 0.0;  // (the re-evaluation changes c for every row: the full refresh runs)
            S.blkDz = 0.0;
            refresh_caches(C, S.hq, S.bEv, S.zg, S.Sp);
            S.hbValid = true;
            S.cDirty = true;
        }
        // rows the pass will have: K - deletes + appends
        int Knew = S.K + (S.appJ >= 0 ? 1 : 0);
#pragma unroll
        for (int t = 0; t < NSL; ++t) Knew -= __popcll(S.del[t]);
        if (S.appAll) {
            int nin = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                nin += __popcll(__ballot(2 * lane + 128 * (k >> 1) < N && st_of(S.Sp, k) == SSQP_IN));
            Knew = nin;
        }
        if (Knew > C.RC) {
            handover = true;
            break;
        }
        const int Kmx = S.K > Knew ? S.K : Knew;
        const bool two = Kmx > 63;           // more than one row slot
        const bool four = NSL > 2 && Kmx > 127;  // ... more than two (the big-factor build)
        // PARK (the 256-register build): the second row slot is in registers only during a pass that has more than 63
        // rows -- under a tenth of the passes of the headline workload; between passes it is parked in the wavefront's
        // global scratch and the registers hold constants, so the one-slot passes are allocated as if it did not exist
        if (PARK && two) slot1_load(S.R, park);
        int step = 0;  // 0: next pass, 1: done, 2: hand over
        do {
            int act;
            if (four) act = wave_sync_factor<NSL>(C, L, S);
            else if (two) act = wave_sync_factor<2>(C, L, S);
            else act = wave_sync_factor<1>(C, L, S);
            if (act == W_BREAK) {
                step = 1;
                break;
            }
            if (act == W_HANDOVER) {
                step = 2;
                break;
            }
            const int K = S.K;
            if (K == 0) {  // ---------------------------------------- freeK!  SSQP.jl:35-59
                // p = V z + q: with no free variable this is the cached hq
                int flag = 0;
                double pa = 0.0;
                unsigned rel = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = 2 * lane + 128 * (k >> 1) + (k & 1);
                    const double pp = (k & 1) ? S.hq[k >> 1].y : S.hq[k >> 1].x;
                    const int s = st_of(S.Sp, k);
                    if (i < N && ((pp >= -tol && s == SSQP_UP) || (pp <= tol && s == SSQP_DN))) {  // :41-47
                        flag = 1;
                        pa = fmax(pa, fabs(pp));
                        rel |= 1u << k;
                    }
                }
                ACCT(C.sBytes += 8ll * N * N + 16ll * N + 4ll * (N + J));
                ACCT(C.sFlops += 2ll * N * N);
                const bool any = __ballot(flag) != 0ull;
                bool done = !any;
                if (any) {
                    const double pm = wave_max(pa);
                    if (pm <= tol) done = true;  // all movable are optimal: statuses restored (:52-55)
                }
                if (done) {
                    if (trace && lane == 0) *trace = ssqp_trace{0, 0, 3, 0};
                    // (no multipliers exist on this exit of the reference: gamma = V z + q, what freeK! tested; lambda = 0)
                    if (C.lamOut && lane < MJ) C.lamOut[lane] = 0.0;
                    if (C.gamOut) {
#pragma unroll
                        for (int m = 0; m < NCH; ++m) {
                            const int r = 2 * lane + 128 * m;
                            if (r < N) *reinterpret_cast<double2 *>(C.gamOut + r) = S.hq[m];
                        }
                    }
                    C.ret = C.iter;  // SSQP.jl:281 (no polishSz! on this exit)
                    step = 1;
                    break;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if ((rel >> k) & 1u) S.Sp &= ~(15u << (4 * k));  // -> IN (code 0)
                S.appAll = true;
                S.hbValid = false;  // variables with z != 0 may have left B
                if (trace && lane == 0) *trace = ssqp_trace{0, 0, 0, 0};
                break;  // (next pass)
            }
            if (K > C.maxK) C.maxK = K;
            if (four) act = wave_pass<NSL>(C, L, S, gscr);
            else if (two) act = wave_pass<2>(C, L, S, gscr);
            else act = wave_pass<1>(C, L, S, gscr);
            step = (act == W_BREAK) ? 1 : 0;
    
        } while (0);
        if (PARK && two) {
            slot1_store(S.R, park);
            slot1_reset(S.R);
        }
        if (step == 2) handover = true;
        if (step != 0) break;
    }
    if (PARK && S.K > 64) slot1_load(S.R, park);  // (the rows are written out below)

    // ---- results: z (free variables from their rows, bound ones from the dense copy / the bounds), S, status
    const bool polished = (C.ret > 0) && (S.K > 0);
    double *zg = S.zg;
    wave_sync();
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const int r = 2 * lane + 128 * m;
        if (r < N) {
            const int s0 = st_of(S.Sp, 2 * m), s1 = st_of(S.Sp, 2 * m + 1);
            if (polished) {  // polishSz!: DN -> d, UP -> u  (SSQP.jl:12-16); the others keep their z
                double2 zz;
                zz.x = __hip_atomic_load(zg + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                zz.y = __hip_atomic_load(zg + r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double2 dd = *reinterpret_cast<const double2 *>(C.dlo + r);
                const double2 uu = *reinterpret_cast<const double2 *>(C.uhi + r);
                zz.x = (s0 == SSQP_DN) ? dd.x : ((s0 == SSQP_UP) ? uu.x : zz.x);
                zz.y = (s1 == SSQP_DN) ? dd.y : ((s1 == SSQP_UP) ? uu.y : zz.y);
                *reinterpret_cast<double2 *>(zg + r) = zz;
            }
            Sg[r] = s0;
            Sg[r + 1] = s1;
        }
    }
    if (lane < J) Sg[N + lane] = ((S.Emask >> lane) & 1u) ? SSQP_EO : SSQP_OE;
    wave_sync();
#pragma unroll
    for (int t = 0; t < NSL; ++t) {
        const int r = lane + KSLOT * t;
        if (r < S.K) zg[S.R.ord[t]] = S.R.zF[t];  // (a variable snapped by polishSz! carries its bound in zF already)
    }
#ifdef SSQP_PHASE_PROFILE
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            (void)__hip_atomic_fetch_add(&g_wphase[k], C.ph[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_add(&g_wphase[32 + k], (unsigned long long)C.pn[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        (void)__hip_atomic_fetch_add(&g_wphase[31], __builtin_amdgcn_s_memtime() - tq0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
    if (lane == 0) {
        if (P.stats) {
            ssqp_stats st;
            const long long done = handover ? C.iter - 1 : (C.iter > P.maxIter ? P.maxIter : C.iter);
            st.iters = done;
            st.alg_bytes = C.sBytes;
            st.read_bytes = C.sRead;
            st.alg_flops = C.sFlops;
            st.sum_k3 = C.sK3;
            st.max_k = C.maxK;
            st.path = 1 | 4 | 16;  // LDS factor, kept-factor engine, wavefront kernel
            if (CAN_RESUME && P.resume) {  // add what the build that handed this QP over counted
                const ssqp_stats s0 = P.stats[prob];
                st.alg_bytes += s0.alg_bytes;
                st.read_bytes += s0.read_bytes;
                st.alg_flops += s0.alg_flops;
                st.sum_k3 += s0.sum_k3;
                st.max_k = st.max_k > s0.max_k ? st.max_k : s0.max_k;
                st.path |= s0.path | 64;  // bit 6: continued in the big-factor build of the wavefront kernel
            }
            P.stats[prob] = st;
        }
        if (handover) {
            P.fbIter[prob] = C.iter - 1;  // passes completed
            const unsigned slot = atomicAdd(P.fbCount, 1u);
            P.fbList[slot] = prob;
        } else {
            P.status[prob] = C.ret;
            if (P.detail) P.detail[prob] = C.det;
        }
    }
    wave_sync();
}

// Three builds of the kernel, one per translation unit (SSQP_WAVE_VARIANT) so that they compile side by side:
//   0: <1, false>  one wavefront per SIMD -- four QPs per CU, 512 registers, up to ~90 free variables, the whole
//                  factor in LDS;
//   1: <2, true>   two per SIMD -- eight QPs per CU, 256 registers, 20 KiB of LDS: rows >= 64 of the factor live in the
//                  wavefront's global scratch (L2), and so does the second row slot between the passes that use it
//                  (PARK).  Half of the QPs of the headline workload end with 64-79 free variables, but only in their
//                  last tenth: carrying the second slot's 36 registers through every pass cost this build some 130
//                  spilled registers and a tenth of its throughput.
//   2: <1, false> with FOUR row slots (NSL = 4) -- the big-factor build: up to 252 free variables, rows >= 64 of the
//                  factor in global scratch (by columns for the forward sweep and the delete, by rows for the back
//                  substitution), every stream through an LDS ring filled by LDS-DMA loads.  It takes the QPs builds
//                  0 / 1 hand over (P.resume: problem ids from their hand-over list, start from the (z, S) they left).
// Each hands a QP that outgrows it over to the next stage (0 / 1 -> 2 -> the workgroup kernel).
// the carve-up of a wavefront's LDS and global scratch (every build)
template <bool PARK>
__device__ __forceinline__ void wave_carve(const SolveParams &P, unsigned char *smem, WLds &L, double *&gscr, double *&park) {
    gscr = P.wscratch + (size_t)blockIdx.x * P.wscratchStride;
    park = gscr;
    double *d0 = reinterpret_cast<double *>(smem);
    const int rc = P.waveRC;
    int o = 0;
    L.F.L0 = d0 + o; o += 2080;
    L.F.LR = nullptr;
    if (NSL > 2) {  // least-squares scratch for up to 256 rows, then rows 64..255 of up to 256 columns
        L.F.L1 = gscr + WAVE_LS_DOUBLES_BIG;
        L.F.R1 = 192;
        L.F.LR = gscr + WAVE_LS_DOUBLES_BIG + 256 * 192 + 64;
    } else if (PARK) {
        L.F.L1 = gscr + WAVE_LS_DOUBLES;
        L.F.R1 = 64;
        park = gscr + WAVE_LS_DOUBLES + 128 * 64;
    } else {
        const int r1 = rc > 64 ? rc - 64 : 0;
        L.F.L1 = d0 + o; o += rc * r1 + 2;
        L.F.R1 = r1 > 0 ? r1 : 1;
    }
    L.H = d0 + o; o += NR * NR;
    L.tr = d0 + o; o += MJX * MJX + 1;
    L.aLrow = d0 + o; o += 16;
    L.yn = d0 + o; o += 16;
    L.GG = d0 + o; o += MJX * MJX + 1;
    L.xn = d0 + o; o += 16;
    L.ra = reinterpret_cast<int16_t *>(d0 + o);
    o += 8;
    L.ring = nullptr;
    L.ringAddr = 0u;
    L.zidx = o;                     // (a double that stays zero: what a masked-out lane of a gather or of the factor sweeps reads)
    L.zero = d0 + o;
    if (threadIdx.x == 0) d0[o] = 0.0;
    o += 2;
    if (NSL > 2) {
        o = (o + 127) / 128 * 128;  // (1 KiB alignment of the DMA pieces)
        L.ring = d0 + o;
        // low half of the flat address of an LDS location = its LDS byte address; read through v_readfirstlane so that
        // the compiler keeps ONE scalar instead of re-deriving it (with its null-pointer select) at every DMA
        L.ringAddr = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(d0 + o));
    }
}

#ifndef SSQP_FULL
template <int WPS, bool PARK, int SLOTS, bool LEAN>  // (SLOTS = NSL, LEAN: part of the kernel's name, so that the builds' kernels differ)
__global__ __launch_bounds__(64, WPS) void ssqp_wave_kernel(SolveParams P) {
    static_assert(LEAN == (SSQP_LEAN_BUILD != 0), "one build per translation unit");
    static_assert(SLOTS == NSL, "one build per translation unit");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WLds L;
    double *gscr, *park;
    wave_carve<PARK>(P, smem, L, gscr, park);
    for (;;) {
        int prob = 0;
        if (threadIdx.x == 0) prob = (int)atomicAdd(P.queue, 1u);
        prob = __builtin_amdgcn_readfirstlane(prob);
        if (CAN_RESUME && P.resume) {  // the QPs another build handed over
            if (prob >= (int)*P.resumeCount) break;
            prob = P.resumeList[prob];
        } else if (prob >= P.nprob) {
            break;
        }
        wave_solve_one<PARK>(P, prob, L, gscr, park);
    }
}
#else
// ---- solveQP(Q) in ONE launch (SSQP.jl:224-234): the wavefront that finds a QP's Phase-1 vertex (ssqp_phase1_wave.h: initQP +
// cDantzigLP, one wavefront per QP) goes straight on into the loop for that QP -- no second launch, no launch tail
// between the stages, the vertex (x0, S0) still L2-hot.  Phase-1's LDS image lies over the loop's (the stages never
// overlap); the loop's one persistent LDS word (the zero word) is set again before the loop starts.  A QP with
// status <= 0 from Phase-1 returns (x0, S, status) as SSQP.jl:230-232 does; a QP Phase-1 does not take (free variables)
// is left on its list for the workgroup Phase-1 kernel, which hands it on like any hand-over.
// The LOOP behind a real CALL: inlined next to Phase-1 (500 registers, every one of them clobbered) the two stages are
// allocated as one function and the loop's hot paths end up with spills they do not have on their own (the pair ran 10 %
// slower than two launches; with Phase-1 as the callee the loop's values that live across the call were reloaded from
// scratch at every use, 50 % slower).  As a callee the loop is allocated by itself, exactly like its stand-alone kernel, and
// the call -- once per QP, nothing of Phase-1 alive across it -- costs the saves of the calling convention.  Arguments of a
// device function travel in vector registers / memory: everything uniform is pinned back to scalars here.
template <class T>
__device__ __forceinline__ T *uni_p(T *p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ size_t uni_sz(size_t v) { return (size_t)uni_p(reinterpret_cast<char *>(v)); }
// (The parameters are read where the kernel itself reads them -- the kernel-argument segment, scalar loads the compiler can
//  re-issue instead of keeping ~45 values alive: as register arguments they cost the callee 5-10 % in scalar spills.)
typedef const __attribute__((address_space(4))) char *karg_t;   // the kernel-argument segment (constant address space)
__device__ __attribute__((noinline)) void loop_call(karg_t kargs_, int prob_) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const karg_t kargs = (karg_t)(unsigned long long)uni_p((const char *)(unsigned long long)kargs_);
    const SolveParams &P = *(const SolveParams *)kargs;   // the kernel's first argument
    WLds L;
    double *gscr, *park;
    wave_carve<false>(P, smem, L, gscr, park);   // (sets the loop's one persistent LDS word: Phase-1's image lay over it)
    wave_sync();
    wave_solve_one<false>(P, uni(prob_), L, gscr, park);
}

// ... and Phase-1 behind a call of its own: the kernel itself is a thin driver with nothing alive across either call, so
// each stage is allocated exactly as in its stand-alone kernel.
template <int NC, int MC>
__device__ __attribute__((noinline)) int phase1_call(karg_t kargs_, int prob_) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const karg_t kargs = (karg_t)(unsigned long long)uni_p((const char *)(unsigned long long)kargs_);
    constexpr size_t qoff = (sizeof(SolveParams) + alignof(p1w::Params) - 1) / alignof(p1w::Params) * alignof(p1w::Params);
    const p1w::Params &Q = *(const p1w::Params *)(kargs + qoff);   // the second argument
    int st = -2;
    const bool took = p1w::solve_one<NC, MC>(Q, uni(prob_), reinterpret_cast<double *>(smem), st);
    return took ? st : -100;
}

template <int NC, int MC>
__global__ __launch_bounds__(64, 1) void ssqp_full_kernel(SolveParams P, p1w::Params Q) {
    static_assert(NSL == 2, "the four-per-CU build");
    const int lane = lane_id();
    // (a callee cannot ask for the kernel-argument segment itself -- the builtin gives it a null pointer: the kernel hands it on)
    const karg_t kargs = (karg_t)__builtin_amdgcn_kernarg_segment_ptr();
    for (;;) {
        int prob = 0;
        if (threadIdx.x == 0) prob = (int)atomicAdd(P.queue, 1u);
        prob = __builtin_amdgcn_readfirstlane(prob);
        if (prob >= P.nprob) break;
        const int st1 = uni(phase1_call<NC, MC>(kargs, prob));
        wave_sync();
        if (st1 == -100) continue;   // (not taken: on the list of the workgroup Phase-1 kernel)
        if (st1 <= 0) {  // SSQP.jl:230-232: return x0, S, status
            const double *x0 = Q.x0 + (size_t)prob * P.N;
            double *z = P.z + (size_t)prob * P.N;
            for (int i = lane; i < P.N; i += 64) z[i] = x0[i];
            if (lane == 0) {
                P.status[prob] = st1;
                if (P.detail) P.detail[prob] = st1 < 0 ? SSQP_DETAIL_SINGULAR_LU : SSQP_DETAIL_NONE;
                if (P.stats) {
                    ssqp_stats z0;
                    z0.iters = 0; z0.alg_bytes = 0; z0.read_bytes = 0; z0.alg_flops = 0; z0.sum_k3 = 0; z0.max_k = 0; z0.path = 128;
                    P.stats[prob] = z0;
                }
            }
            continue;
        }
        loop_call(kargs, prob);
    }
}
#endif

}  // namespace

#ifdef SSQP_FULL
template <int NC, int MC>
static hipError_t launch_full_t(const SolveParams &P, const p1w::Params &Q, int grid, size_t lds, hipStream_t stream) {
    static unsigned long long ldsSet = 0ull;
    hipError_t e = allow_full_lds(reinterpret_cast<const void *>(&ssqp_full_kernel<NC, MC>), &ldsSet);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((ssqp_full_kernel<NC, MC>), dim3(grid), dim3(64), lds, stream, P, Q);
    return hipGetLastError();
}
hipError_t launch_solve_full(const SolveParams &P, int grid, const double *A, const double *G, const double *b, const double *g,
                             double tolLP, double *x0, int32_t *p1status, unsigned int *p1Count, int *p1List, hipStream_t stream) {
    p1w::Params Q;
    Q.nprob = P.nprob; Q.N = P.N; Q.M = P.M; Q.J = P.J;
    Q.A = A; Q.G = G; Q.b = b; Q.g = g; Q.d = P.d; Q.u = P.u;
    Q.tol = tolLP;
    Q.x0 = x0; Q.S = P.S; Q.status = p1status;
    Q.fbCount = p1Count; Q.fbList = p1List;
    const int M0 = P.M + P.J, N1 = P.N + P.J + M0;
    // (three builds of the pair: the loop's code is in every one of them)
    if (M0 <= 4) {
        if (N1 <= 64 * 5) {
            const size_t l1 = (size_t)p1w::lds_doubles<5, 4>() * 8, l = l1 > (size_t)P.waveLdsBytes ? l1 : (size_t)P.waveLdsBytes;
            return launch_full_t<5, 4>(P, Q, grid, l, stream);
        }
        const size_t l1 = (size_t)p1w::lds_doubles<9, 4>() * 8, l = l1 > (size_t)P.waveLdsBytes ? l1 : (size_t)P.waveLdsBytes;
        return launch_full_t<9, 4>(P, Q, grid, l, stream);
    }
    const size_t l1 = (size_t)p1w::lds_doubles<9, 11>() * 8, l = l1 > (size_t)P.waveLdsBytes ? l1 : (size_t)P.waveLdsBytes;
    return launch_full_t<9, 11>(P, Q, grid, l, stream);
}
}  // namespace ssqp
#else
#if SSQP_LEAN_BUILD
#define WV_NAME(base) base##_lean
#define WV_LEANARG true
#else
#define WV_NAME(base) base
#define WV_LEANARG false
#endif
#if SSQP_WAVE_VARIANT == 0
#define WV_KERNEL ssqp_wave_kernel<1, false, 2, WV_LEANARG>
#define WV_LAUNCH WV_NAME(launch_wave_v0)
#define WV_PHASES WV_NAME(wave_phases_v0)
#elif SSQP_WAVE_VARIANT == 1
#define WV_KERNEL ssqp_wave_kernel<2, true, 2, WV_LEANARG>
#define WV_LAUNCH WV_NAME(launch_wave_v1)
#define WV_PHASES WV_NAME(wave_phases_v1)
#elif SSQP_WAVE_VARIANT == 2
#define WV_KERNEL ssqp_wave_kernel<1, false, 4, WV_LEANARG>
#define WV_LAUNCH WV_NAME(launch_wave_v2)
#define WV_PHASES WV_NAME(wave_phases_v2)
#else
#error "SSQP_WAVE_VARIANT: 0, 1 or 2"
#endif

hipError_t WV_LAUNCH(const SolveParams &P, int grid, hipStream_t stream) {
    static unsigned long long ldsSet = 0ull;
    hipError_t e = allow_full_lds(reinterpret_cast<const void *>(&WV_KERNEL), &ldsSet);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(WV_KERNEL, dim3(grid), dim3(64), (size_t)P.waveLdsBytes, stream, P);
    return hipGetLastError();
}
#ifdef SSQP_PHASE_PROFILE
int WV_PHASES(unsigned long long *out64, int reset) {  // adds this build's stamps to out64
    unsigned long long h[64];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wphase), sizeof(h)) != hipSuccess) return 1;
    for (int k = 0; k < 64; ++k) out64[k] += h[k];
    if (reset) {
        static unsigned long long zero[64];
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_wphase), zero, sizeof(zero)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

#if SSQP_WAVE_VARIANT == 0 && !SSQP_LEAN_BUILD
hipError_t launch_wave_v1(const SolveParams &P, int grid, hipStream_t stream);
hipError_t launch_wave_v2(const SolveParams &P, int grid, hipStream_t stream);
hipError_t launch_wave_v0_lean(const SolveParams &P, int grid, hipStream_t stream);
hipError_t launch_wave_v1_lean(const SolveParams &P, int grid, hipStream_t stream);
hipError_t launch_wave_v2_lean(const SolveParams &P, int grid, hipStream_t stream);
bool wave_kernel_applies(int N, int M, int J) {
    return (N % 2 == 0) && N >= 2 && N <= WAVE_MAXN && (M + J) <= WAVE_MJ;
}
int wave_lds_bytes(int rc) {  // rc <= 0: the builds that keep rows >= 64 in global scratch (eight-per-CU, big-factor)
    const int r1 = rc > 64 ? rc - 64 : 0;
    const int l1 = rc > 0 ? rc * r1 + 2 : 0;
    const int dbl = 2080 + l1 + NR * NR + 2 * (MJX * MJX + 1) + 16 * 3 + 8 + 2;  // (+ the zero word)
    return dbl * 8;
}
int wave_lds_bytes_big() {  // big-factor build: the same without LDS rows >= 64, plus the column ring (1 KiB aligned)
    return (wave_lds_bytes(0) + 1023) / 1024 * 1024 + RING_BYTES + 2048;   // (+ 2 KiB: the screening pass's dense fp32 vectors)
}
size_t wave_scratch_doubles(int variant) {
    // least-squares scratch, then (eight-per-CU build) rows 64..127 of up to 128 columns of the factor and the parked
    // second row slot; (big-factor build) rows 64..255 of up to 256 columns
    // ... and V rounded to fp32 (N <= 256: 256 x 256 floats) plus one DMA piece of slack behind it
    if (variant == 2) return (size_t)WAVE_LS_DOUBLES_BIG + 256 * 192 + 64 + 192 * LR_STRIDE + 64 + 256 * 256 / 2 + 256;
    return (size_t)WAVE_LS_DOUBLES + 128 * 64 + PARK_FIELDS * 64 + 64;
}
hipError_t launch_solve_wave(const SolveParams &P, int grid, int variant, hipStream_t stream) {
    // a launch that asks for neither statistics nor a trace gets the builds that do not carry them (-DSSQP_WAVE_LEAN)
    if (!P.stats && !P.trace) {
        if (variant == 2) return launch_wave_v2_lean(P, grid, stream);
        if (variant == 1) return launch_wave_v1_lean(P, grid, stream);
        return launch_wave_v0_lean(P, grid, stream);
    }
    if (variant == 2) return launch_wave_v2(P, grid, stream);
    if (variant == 1) return launch_wave_v1(P, grid, stream);
    return launch_wave_v0(P, grid, stream);
}
#endif  // SSQP_WAVE_VARIANT == 0 && !SSQP_LEAN_BUILD

}  // namespace ssqp

#if defined(SSQP_PHASE_PROFILE) && SSQP_WAVE_VARIANT == 0 && !SSQP_LEAN_BUILD
namespace ssqp {
int wave_phases_v1(unsigned long long *out64, int reset);
int wave_phases_v2(unsigned long long *out64, int reset);
}
extern "C" int ssqp_debug_wave_phases(unsigned long long *out64, int reset) {
    for (int k = 0; k < 64; ++k) out64[k] = 0;
    return ssqp::wave_phases_v0(out64, reset) | ssqp::wave_phases_v1(out64, reset) | ssqp::wave_phases_v2(out64, reset);
}
#endif
#endif  // SSQP_FULL
