// ssqp_phase1_wave.h -- gfx950: Phase-1 of solveQP(Q) with ONE 64-lane WAVEFRONT per QP (device code, included by
// ssqp_phase1_wave.hip -- the stand-alone kernel -- and by ssqp_wave.hip -- the single-launch solveQP(Q)).
//
// What it replaces: initQP (reference: src/SSQP.jl:461-560) and the bounded-variable simplex it calls, cDantzigLP
// (src/Simplex.jl:445-615), exactly as ssqp_host.cpp (phase1_one / BoundedSimplex) and the workgroup kernel
// ssqp_phase1.hip do, and BIT-IDENTICAL to both: every decision (largest-distance Dantzig pricing with the switch to
// Bland's rule after N1 loops, first-minimum ratio test, bound flips, sorted basis, inv(lu(A[:,B])) with partial
// pivoting) and every rounding is the host's -- each sum runs in the host's order with separately rounded multiply and
// add (`#pragma clang fp contract(off)` in every function here), IEEE division and square root.
//
// Why a wavefront: the workgroup kernel is a chain of ~10 barrier-separated steps per simplex pass in which three of
// four wavefronts mostly wait (42 s_barrier in its code object, 81 % -> 50 % of its wave cycles parked).  Here
//   * lane l OWNS the columns k = l, l + 64, ... of the LP [A; G | slacks | artificials] (N1 = N + J + M0 <= 64 NC of them):
//     their entries (M0 <= MC doubles each), reduced-cost dot product, value x_k, norm, status bits live in ITS registers
//     -- the LP matrix is never re-read from memory;
//   * lane r < M0 also owns ROW r of the basis: row r of inv(B), xb_r, basis[r], the bounds of that basic variable, and
//     its column of the basis matrix;
//   * everything uniform (the entering column, a row of inv(B) during the refresh of Y, pivots of the LU) travels by
//     v_readlane into scalar registers and is used as a scalar operand -- no LDS round trip in any dependent chain, no
//     barrier anywhere; LDS only holds the bounds (looked up by variable id) and the two small transposes;
//   * inv(lu(B)): the elimination broadcasts the pivot column by v_readlane, every lane updates its own column; the
//     columns of the inverse take one lane each (operations per element and their order as in the host's invert_lu).
// A QP with a free variable (u = +Inf and d = -Inf: initQP splits it into two columns, SSQP.jl:484-505) is not taken: it
// is appended to a list that the workgroup kernel then works off (ssqp_api.hip).
#ifndef SSQP_PHASE1_WAVE_H
#define SSQP_PHASE1_WAVE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssqp_hip.h"
#include "ssqp_device.h"

namespace ssqp {
namespace p1w {

constexpr double INF = __builtin_huge_val();

// ---- diagnostic build only (-DSSQP_PHASE_PROFILE): cycles per phase, accumulated in registers, flushed once per QP ----
#ifdef SSQP_PHASE_PROFILE
static __device__ unsigned long long g_p1wphase[16];
#define W1_DECL unsigned long long w1t = __builtin_amdgcn_s_memtime(); unsigned long long w1a[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define W1_STAMP(slot)                                                   \
    do {                                                                 \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();      \
        w1a[slot] += t_ - w1t;                                           \
        w1t = __builtin_amdgcn_s_memtime();                              \
    } while (0)
#define W1_COUNT(slot) w1a[slot] += 1
#define W1_FLUSH()                                                                                                     \
    do {                                                                                                               \
        if ((threadIdx.x & 63) == 0)                                                                                   \
            for (int k_ = 0; k_ < 16; ++k_)                                                                            \
                (void)__hip_atomic_fetch_add(&g_p1wphase[k_], w1a[k_], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    \
    } while (0)
#else
#define W1_DECL do { } while (0)
#define W1_STAMP(slot) do { } while (0)
#define W1_COUNT(slot) do { } while (0)
#define W1_FLUSH() do { } while (0)
#endif

struct Params {
    int nprob, N, M, J;
    const double *A, *G, *b, *g, *d, *u;   // per problem, back to back (A: M x N, G: J x N, column-major)
    double tol;
    double *x0;
    int32_t *S;
    int32_t *status;
    unsigned int *fbCount;   // QPs this kernel does not take (free variables): count and list, for the workgroup kernel
    int *fbList;
};

// LDS of one wavefront, in doubles: lo, hi, x, ||A[:,k]|| by variable id (64 NC each), the basis matrix by columns and the MC x MC
// transposes, the cached terms of the xb sum (TCAP slots of MC), MC scratch, then ints
template <int NC, int MC>
__host__ __device__ constexpr int lds_doubles() {
    return 4 * 64 * NC + 2 * MC * MC + 64 * MC + 64 * ((MC + 2) & ~1) + 2 * MC + 2 * MC + 32;  // (64 = TCAP; the tail: ints)
}

template <int NC>
__device__ __forceinline__ double col_get(const double (&v)[NC], int k) {  // value of column k (uniform k)
    double x = 0.0;
    const int l = k & 63, c = k >> 6;
#pragma unroll
    for (int t = 0; t < NC; ++t)
        if (c == t) x = readlane_f64(v[t], l);
    return x;
}
template <int NC>
__device__ __forceinline__ void col_set(double (&v)[NC], int k, double x) {  // uniform k
    const bool mine = (int)(threadIdx.x & 63) == (k & 63);
    const int c = k >> 6;
#pragma unroll
    for (int t = 0; t < NC; ++t) v[t] = (mine && c == t) ? x : v[t];
}

// inv(lu(B)) for the n x n basis matrix whose column j is in lane j's `col` (n <= MC), partial pivoting, the host's
// operations per element and their order (ssqp_host.cpp invert_lu).  On return lane j holds COLUMN j of the inverse in
// `col`.  Returns false (to every lane) when a pivot is exactly 0 (lu() of the reference throws, Simplex.jl:590).
// The elimination never leaves the registers: step k reads column k out of lane k by v_readlane -- the values are then
// uniform, so the pivot search, the reciprocal and the scaling run once for the wavefront, the row swap is a uniform
// branch -- and every lane right of k updates its own column with scalar operands.  The substitutions of column c of
// the inverse touch only x_c: lane c runs both, taking L(i,k) / U(i,k) from lane k the same way.
template <int MC>
__device__ __forceinline__ bool invert_lu_regs(double (&col)[MC], int n) {
#pragma clang fp contract(off)
    // Straight-line code: every step runs for every k < MC with selects instead of branches (a branch per step makes the
    // register allocator re-shuffle the whole column at every join -- a third of the instructions of the first version).
    // Rows and lanes n .. MC - 1 are ZERO (the LP's columns are zero-padded; the lanes are cleared here): a padded row or
    // lane takes part with zeros, which change no bit of a real entry, and a padded step has its reciprocal forced to 0.
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < MC; ++i) col[i] = (lane < n) ? col[i] : 0.0;
    int pv[MC];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < MC; ++k) {
        const bool step = k < n;  // uniform
        // the FIRST largest |a(i, k)|, i = k .. n - 1 (the host's strict ">" scan): every lane scans its own column, lane k's
        // verdict counts (per-lane arithmetic: the scalar unit has no f64 compare)
        int pl = k;
        double best = fabs(col[k]);
#pragma unroll
        for (int i = k + 1; i < MC; ++i) {
            const double v = fabs(col[i]);
            const bool gt = v > best;
            best = gt ? v : best;
            pl = gt ? i : pl;
        }
        const int p = __builtin_amdgcn_readlane(pl, k);   // (a padded step: every entry is zero, so p = k)
        best = readlane_f64(best, k);
        ok = ok && (!step || best != 0.0);
        pv[k] = p;
        {   // rows k and p change places in every column (p == k: nothing matches)
            const double ak = col[k];
            double ap = ak;
#pragma unroll
            for (int i = k + 1; i < MC; ++i) {
                ap = (i == p) ? col[i] : ap;
                col[i] = (i == p) ? ak : col[i];
            }
            col[k] = ap;
        }
        const double rq = 1.0 / readlane_f64(col[k], k);
        const double r = step ? rq : 0.0;
        const double akj = col[k];
#pragma unroll
        for (int i = k + 1; i < MC; ++i) {
            const double li = readlane_f64(col[i], k) * r;           // a(i, k) *= r
            const double upd = col[i] - li * akj;                    // a(i, j) -= a(i, k) * a(k, j)   (j > k)
            col[i] = (lane > k) ? upd : ((lane == k) ? li : col[i]);
        }
    }
    if (!ok) return false;
    // column `lane` of the inverse: L U x = P e_lane.  P e_c: the host applies the row swaps to e_c in order; the permuted
    // unit vector has its 1 where that sequence of swaps sends index c
    int pos = lane;
#pragma unroll
    for (int k = 0; k < MC; ++k) pos = (pos == k) ? pv[k] : ((pos == pv[k]) ? k : pos);   // (pv[k] == k: no change)
    double xc[MC];
#pragma unroll
    for (int i = 0; i < MC; ++i) xc[i] = (i == pos) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < MC; ++k) {  // forward: x[i] -= L(i, k) x[k]  for i > k
#pragma unroll
        for (int i = k + 1; i < MC; ++i) xc[i] -= readlane_f64(col[i], k) * xc[k];
    }
#pragma unroll
    for (int kk = 0; kk < MC; ++kk) {  // backward: x[k] /= U(k, k), then x[i] -= U(i, k) x[k]  for i < k
        const int k = MC - 1 - kk;
        const double q = xc[k] / readlane_f64(col[k], k);
        xc[k] = (k < n) ? q : 0.0;
#pragma unroll
        for (int i = 0; i < MC; ++i)
            if (i < k) xc[i] -= readlane_f64(col[i], k) * xc[k];
    }
#pragma unroll
    for (int i = 0; i < MC; ++i) col[i] = xc[i];
    return true;
}

// One QP.  `lds`: lds_doubles<NC, MC>() doubles of this wavefront's LDS.  Returns false when the QP was not taken (it has
// free variables and now sits on P.fbList); otherwise x0, S and status are written (statusOut: the same status, uniform).
// Padding: rows M0 .. MC - 1 of every column, of inv(B) and of the row vectors are exactly +0.0, so the loops over rows
// carry no guard on M0: a padded row adds 0.0 * 0.0 = +0.0 to a sum that started at +0.0 and therefore is never -0.0 (x + y
// = -0.0 needs x = y = -0.0) -- no bit of any sum changes.
constexpr int TCAP = 64;
#ifndef RGBIG
#define RGBIG 1
#endif   // nonbasic columns at a nonzero value whose xb terms are cached (more: recomputed every pass)
template <int NC, int MC>
__device__ __forceinline__ bool solve_one(const Params &P, int prob, double *lds, int &statusOut) {
#pragma clang fp contract(off)
    statusOut = -2;
    const int lane = threadIdx.x & 63;
    W1_DECL;
    const int N = P.N, M = P.M, J = P.J, M0 = M + J;
    const int N0 = N + J, N1 = N0 + M0;
    const double *A = P.A + (size_t)prob * M * N;
    const double *G = P.G + (size_t)prob * J * N;
    const double *b = P.b + (size_t)prob * M;
    const double *g = P.g + (size_t)prob * J;
    const double *d = P.d + (size_t)prob * N;
    const double *u = P.u + (size_t)prob * N;
    const double tol = P.tol;
    double *x0 = P.x0 + (size_t)prob * N;
    int32_t *S = P.S + (size_t)prob * (N + J);

    // by variable id k (lane k & 63 owns it): bounds, value, column norm -- what a pass touches once at most stays out of
    // the register file, which the LP's columns fill
    double *lo = lds, *hi = lo + 64 * NC, *xL = hi + 64 * NC, *nL = xL + 64 * NC;
    double *T = nL + 64 * NC;            // MC x MC: columns on their way between lanes
    double *Bc = T + MC * MC;            // MC x MC: column j of the basis matrix A1[:, basis[j]] at Bc[j * MC ..]
    double *tl = Bc + MC * MC;           // TCAP x MC: cached terms Y[:,k] x_k of the xb sum, one slot per listed column
    constexpr int MCP = (MC + 2) & ~1;   // a listed column in LDS: its MC entries, then x_k (16-byte rows)
    double *AL = tl + TCAP * MC;         // TCAP x MCP: the listed columns side by side while their terms are formed
    double *sg = AL + TCAP * MCP;        // MC: signs of the artificial columns
    int *ib = reinterpret_cast<int *>(sg + 2 * MC);  // 2 MC ints
    int *ib2 = ib + 2 * MC;              // TCAP ints: column ids of the list

    // ---- the LP of initQP (SSQP.jl:484-526): columns [A; G | slack | artificials], this lane's columns in registers
    double a[NC][MC], sd[NC];
    // per column slot: status (2 bits: IN 0, DN 1, UP 2), nonbasic, (-inf, u] variable, value x != 0
    unsigned stw = 0, nbw = 0, upw = 0, nzw = 0;
    bool anyFree = false;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = lane + 64 * c;
        const int ks = k < N ? k : 0;
        const double dk = d[ks], uk = u[ks];
        const bool structural = k < N;
        const bool noUp = uk == INF, noLo = dk == -INF;
        anyFree = anyFree || (structural && noUp && noLo);
        const bool upOnly = structural && noLo && !noUp;
        double lok = 0.0, hik = INF;
        if (structural) {
            lok = upOnly ? -uk : dk;   // (-inf, u] variables are sign-flipped (SSQP.jl:506-509)
            hik = upOnly ? INF : uk;
        }
        lo[k] = lok;
        hi[k] = hik;
        if (upOnly) upw |= 1u << c;
#pragma unroll
        for (int r = 0; r < MC; ++r) {
            double v = 0.0;
            if (r < M0) {  // uniform
                if (structural) {
                    v = (r < M) ? A[(size_t)ks * M + (r < M ? r : 0)] : G[(size_t)ks * J + (r >= M ? r - M : 0)];
                    v = upOnly ? -v : v;
                } else if (k < N0) {
                    v = (r == M + (k - N)) ? 1.0 : 0.0;
                }
            }
            a[c][r] = v;
        }
        stw |= 1u << (2 * c);                 // DN
        if (k < N0) nbw |= 1u << c;
        xL[k] = lok;                          // x[k] = lo[k]  (every nonbasic starts at its lower bound)
        if (lok != 0.0) nzw |= 1u << c;
        sd[c] = 0.0;
    }
    if (__ballot(anyFree) != 0ull) {
        if (lane == 0) {
            const unsigned slot = atomicAdd(P.fbCount, 1u);
            P.fbList[slot] = prob;
        }
        return false;
    }
    wave_sync();
    // rows: lane r < M0 owns row r of the basis
    const int rr = lane < M0 ? lane : 0;
    double rhs = 0.0;
    if (lane < M0) rhs = (lane < M) ? b[lane < M ? lane : 0] : g[lane >= M ? lane - M : 0];
    // start = sum over the columns k < N0 with lo != 0, ascending, of A1[:,k] * lo[k]   (SSQP.jl:511-526)
    double start = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = lane + 64 * c;
        unsigned long long m = __ballot(k < N0 && ((nzw >> c) & 1u));
        while (m) {  // uniform
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const double lk = xL[l + 64 * c];
            double ar = 0.0;
#pragma unroll
            for (int t = 0; t < MC; ++t) {
                const double v = readlane_f64(a[c][t], l);
                ar = (lane == t) ? v : ar;
            }
            start += ar * lk;
        }
    }
    const double sgn = rhs >= start ? 1.0 : -1.0;
    if (lane < MC) sg[lane] = sgn;
    wave_sync();
    // rows of inv(B): lane r < M0 owns row r; the lanes r + M0 g (g < RGRP) hold copies, so that the terms of the xb sum --
    // M0 rows x the listed columns -- are formed by RGRP M0 lanes at once instead of M0
    const int RGRP = (M0 * 16 <= 64) ? 16 : 64 / M0;   // (M0 <= 11: at least 5 groups)
    const int rgrp = lane / M0, rrow = lane - rgrp * M0;
    const bool rlane = rgrp < RGRP;
    double ivr[MC];
#pragma unroll
    for (int t = 0; t < MC; ++t) {
        ivr[t] = (rlane && rrow == t) ? sg[rrow] : 0.0;   // invB = diag(sgn)
        if (lane < MC) Bc[lane * MC + t] = (lane == t && lane < M0) ? sgn : 0.0;   // column `lane` of the basis matrix: the artificial column
    }
    double xb = (lane < M0) ? fabs(start - rhs) : 0.0;
    int bas = N0 + rr;
    double blo = 0.0, bhi = INF;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = lane + 64 * c;
        if (k >= N0 && k < N1) {  // artificial column j = k - N0: sgn_j e_j, basic (IN)
            const int j = k - N0;
            const double sj = sg[j];
#pragma unroll
            for (int r = 0; r < MC; ++r) a[c][r] = (r == j) ? sj : 0.0;
            stw &= ~(3u << (2 * c));
        }
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < MC; ++r) s += a[c][r] * a[c][r];
        nL[k] = sqrt(s);
    }

    // Y = invB * A[:, nonbasic] is never stored (as in ssqp_phase1.hip): the pricing needs Y[:,k] . c[basis], formed
    // here for every column this lane owns, rows in the host's order; row r of invB comes out of lane r as scalars.
    // c[basis[r]] is 1 for an artificial variable and 0 otherwise.  A row with c = 0 adds s * 0.0 = +-0.0 to a sum that
    // started at +0.0 and is never -0.0: it changes no bit of it and is skipped -- more than half of the rows, on average
    // (only a row sum that is Inf / NaN would have left a NaN behind).  The rows that count go up to three at a time, so that
    // an entry of the LP (half of them live in the accumulator half of the register file) is fetched once per group.
    constexpr int RG = (NC * MC > 60) ? RGBIG : 3;   // rows per round (the accumulators of a round: RG x NC doubles)
    auto refreshY = [&]() __attribute__((always_inline)) {
#pragma clang fp contract(off)
#pragma unroll
        for (int c = 0; c < NC; ++c) sd[c] = 0.0;
        unsigned long long rows = __ballot(lane < M0 && bas >= N0);
        while (rows) {  // uniform: up to three artificial rows per round, ascending
            int r3[RG];
            bool on[RG];
#pragma unroll
            for (int q = 0; q < RG; ++q) {
                on[q] = rows != 0ull;
                r3[q] = on[q] ? __ffsll((long long)rows) - 1 : 0;
                rows &= rows - 1;   // (0 stays 0)
            }
            double s[RG][NC];
#pragma unroll
            for (int q = 0; q < RG; ++q)
#pragma unroll
                for (int c = 0; c < NC; ++c) s[q][c] = 0.0;
#pragma unroll
            for (int t = 0; t < MC; ++t) {
                double iv[RG];
#pragma unroll
                for (int q = 0; q < RG; ++q) iv[q] = readlane_f64(ivr[t], r3[q]);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const double act = a[c][t];
#pragma unroll
                    for (int q = 0; q < RG; ++q) s[q][c] += iv[q] * act;
                }
            }
#pragma unroll
            for (int q = 0; q < RG; ++q) {
                const double cb = on[q] ? 1.0 : 0.0;   // (a round's unused places repeat row 0 with weight 0: + +-0.0)
#pragma unroll
                for (int c = 0; c < NC; ++c) sd[c] += s[q][c] * cb;
            }
        }
    };
    refreshY();
    W1_STAMP(0);  // set-up

    // ---- the xb sum's terms Y[:,k] x_k of the nonbasic columns at a nonzero value, cached between basis changes: lane p
    // holds the column id (kkv) and the LDS slot (ordv) of the p-th listed column in ascending order, lane r reads
    // its row of a slot.  A bound flip touches one term; the sum is re-added in order every pass.
    int kkv = 0x7fffffff, ordv = 0, cnt = 0;
    unsigned long long used = 0ull;
    bool rebuild = true, cached = false;

    int status = 1;
    long loop = 0;
    for (;;) {
        loop += 1;
        // (the reference's loop has no limit, Simplex.jl:486: an LP still pivoting after 64 N1 + 1024 passes -- NaN-poisoned or
        //  cycling in floating point -- is given up as a numerical error, as in ssqp_phase1.hip)
        if (loop > 64l * N1 + 1024) {
            status = -1;
            break;
        }
        W1_COUNT(14);
        const bool bland = loop > N1;
        // ---- price: signed reduced costs, the entering candidate (first maximum of h / ||A[:,k]||; Bland: first candidate)
        double best = -INF;
        int bidx = 0x7fffffff;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = lane + 64 * c;
            const double ck = k >= N0 ? 1.0 : 0.0;  // (= cost[k]: the Phase-1 objective is the sum of the artificials)
            double hv = ck - sd[c];
            hv = (((stw >> (2 * c)) & 3u) == (unsigned)SSQP_DN) ? -hv : hv;
            const bool candk = ((nbw >> c) & 1u) && hv > tol;
            double v = hv / nL[k];
            v = bland ? 0.0 : v;
            const bool take = candk && v > best;   // (ascending k inside the lane: the first maximum stays)
            best = take ? v : best;
            bidx = take ? k : bidx;
        }
        {
            const KeyMin km = wave_keymin(KeyMin{-best, bidx});
            bidx = km.ord;
        }
        W1_STAMP(1);  // pricing + first maximum
        if (bidx == 0x7fffffff) break;  // no improving candidate: optimal
        const int k = bidx, lk = k & 63, ck_ = k >> 6;
        // ---- the entering column, as scalars; p = invB * A[:,k]: lane r forms row r
        double ak[MC];
#pragma unroll
        for (int t = 0; t < MC; ++t) ak[t] = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (ck_ == c) {
#pragma unroll
                for (int t = 0; t < MC; ++t) ak[t] = readlane_f64(a[c][t], lk);
            }
        double p = 0.0;
#pragma unroll
        for (int t = 0; t < MC; ++t) p += ivr[t] * ak[t];
        const bool fromLower = ((((unsigned)__builtin_amdgcn_readlane((int)stw, lk)) >> (2 * ck_)) & 3u) == (unsigned)SSQP_DN;
        const double loK = lo[k], hiK = hi[k];
        const double rangeK = hiK - loK;
        // ---- ratio test (Simplex.jl:499-569): first minimum (entering from below) / first maximum over the basic rows
        const bool pos = p > tol, neg = p < -tol;
        const bool cand = (lane < M0) && (pos || neg);
        const bool toLower = fromLower ? pos : neg;
        const double ratio = cand ? (xb - (toLower ? blo : bhi)) / p : 0.0;
        const bool any = __ballot(cand) != 0ull;
        // (candidates first: a candidate whose ratio is +-inf still beats every row that is none)
        const KeyMin kr = wave_keymin(KeyMin{cand ? (fromLower ? ratio : -ratio) : INF, cand ? lane : lane + 64});
        const int lrow = kr.ord & 63;
        const double lr = fromLower ? kr.v : -kr.v;
        const int lto = __builtin_amdgcn_readlane(toLower ? SSQP_DN : SSQP_UP, lrow);
        int action = 0, leaveStatus = SSQP_DN;  // > 0: basis row + 1 leaves, -1 flip to UP, -2 flip to DN
        bool unbounded = false;
        if (fromLower) {
            const bool finiteUp = hiK < INF;
            if (!any) {
                if (!finiteUp) unbounded = true;
                else action = -1;
            } else if (finiteUp && lr >= rangeK) {
                action = -1;
            } else if (!finiteUp && isinf(lr)) {
                unbounded = true;
            } else {
                action = lrow + 1;
                leaveStatus = lto;
            }
        } else {
            if (!any) action = -2;
            else if (lr <= -rangeK) action = -2;
            else action = lrow + 1, leaveStatus = lto;
        }
        W1_STAMP(2);  // entering column + ratio test
        if (unbounded) {
            status = 3;
            break;
        }
        if (action < 0) {  // bound flip of the entering variable
            const unsigned ns = action == -1 ? (unsigned)SSQP_UP : (unsigned)SSQP_DN;
            const double xnew = action == -1 ? hiK : loK;
            const double xold = xL[k];
            if (lane == lk) {
                stw = (stw & ~(3u << (2 * ck_))) | (ns << (2 * ck_));
                nzw = (nzw & ~(1u << ck_)) | ((xnew != 0.0 ? 1u : 0u) << ck_);
                xL[k] = xnew;
            }
            if (cached && !rebuild) {  // the one term of the cached xb sum that changes: Y[:,k] = p, the value just formed
                const bool was = xold != 0.0, is = xnew != 0.0;
                if (was) {  // (uniform) in the list: its place and slot
                    const int at = __ffsll((long long)__ballot(lane < cnt && kkv == k)) - 1;
                    const int slot = __builtin_amdgcn_readlane(ordv, at);
                    if (is) {
                        if (lane < MC) tl[slot * MC + lane] = p * xnew;
                    } else {   // leaves the list: the later entries move down
                        used &= ~(1ull << slot);
                        const int kn = __builtin_amdgcn_update_dpp(0x7fffffff, kkv, 0x130, 0xF, 0xF, false);
                        const int on = __builtin_amdgcn_update_dpp(0, ordv, 0x130, 0xF, 0xF, false);
                        kkv = (lane >= at) ? kn : kkv;
                        ordv = (lane >= at) ? on : ordv;
                        cnt -= 1;
                    }
                } else if (is) {
                    if (cnt >= TCAP) {
                        cached = false;   // (no room: this QP goes back to recomputing the terms every pass)
                    } else {
                        const int at = __popcll(__ballot(lane < cnt && kkv < k));
                        const int slot = __ffsll((long long)~used) - 1;
                        used |= 1ull << slot;
                        if (lane < MC) tl[slot * MC + lane] = p * xnew;
                        const int kp = __builtin_amdgcn_update_dpp(0x7fffffff, kkv, 0x138, 0xF, 0xF, false);
                        const int op = __builtin_amdgcn_update_dpp(0, ordv, 0x138, 0xF, 0xF, false);
                        kkv = (lane > at) ? kp : ((lane == at) ? k : kkv);
                        ordv = (lane > at) ? op : ((lane == at) ? slot : ordv);
                        cnt += 1;
                    }
                }
                wave_sync();
            }
        } else {
            const int row = action - 1;
            const int leaving = __builtin_amdgcn_readlane(bas, row);
            const double newx = readlane_f64(leaveStatus == SSQP_DN ? blo : bhi, row);  // the bound the leaving variable goes to
            // basis[row] = k, then sort(basis): lane j's new place is the number of smaller entries; the columns of the
            // basis matrix (kept in LDS between pivots) move with them
            double col[MC];
#pragma unroll
            for (int t = 0; t < MC; ++t) col[t] = (lane == row) ? ak[t] : Bc[rr * MC + t];
            if (lane == row) bas = k;
            int rank = 0;
#pragma unroll
            for (int i = 0; i < MC; ++i) rank += (i < M0 && __builtin_amdgcn_readlane(bas, i) < bas) ? 1 : 0;
            if (lane < M0) {
                ib[rank] = bas;
#pragma unroll
                for (int t = 0; t < MC; ++t) T[rank * MC + t] = col[t];
            }
            wave_sync();
            bas = ib[rr];
#pragma unroll
            for (int t = 0; t < MC; ++t) col[t] = T[rr * MC + t];
            if (lane < M0) {
#pragma unroll
                for (int t = 0; t < MC; ++t) Bc[lane * MC + t] = col[t];
            }
            blo = lo[bas];
            bhi = hi[bas];
            wave_sync();
            W1_STAMP(3);  // basis sort + columns
            W1_COUNT(15);
            if (!invert_lu_regs<MC>(col, M0)) {  // lu() of the reference throws (Simplex.jl:590)
                status = -1;
                break;
            }
            W1_STAMP(4);  // inv(lu(B))
            // lane j holds column j of the inverse; row r goes to lane r
            if (lane < M0) {
#pragma unroll
                for (int t = 0; t < MC; ++t) T[t * MC + lane] = col[t];   // T[i][j] = inv(i, j)
            }
            wave_sync();
#pragma unroll
            for (int t = 0; t < MC; ++t) ivr[t] = (rlane && t < M0) ? T[rrow * MC + t] : 0.0;
            wave_sync();
            // statuses and values: S[k] = IN, S[leaving] = leaveStatus, x[leaving] = its bound
            {
                const int ll = leaving & 63, lc = leaving >> 6;
                if (lane == lk) {
                    stw &= ~(3u << (2 * ck_));
                    nbw &= ~(1u << ck_);
                }
                if (lane == ll) {
                    stw = (stw & ~(3u << (2 * lc))) | ((unsigned)leaveStatus << (2 * lc));
                    nbw |= 1u << lc;
                    nzw = (nzw & ~(1u << lc)) | ((newx != 0.0 ? 1u : 0u) << lc);
                    xL[leaving] = newx;
                }
            }
            W1_STAMP(5);  // rows of the inverse, statuses
            refreshY();
            rebuild = true;
            W1_STAMP(6);  // Y . c
        }
        // ---- xb = invB * b - Y * x[nonbasic]: the nonbasic columns at a nonzero value, ascending, one rounded multiply and
        // one rounded add per term (Simplex.jl:599)
        W1_STAMP(8);  // bound flip + the cached term it changes
        double a2 = 0.0;
        if (rebuild || !cached) {  // every term anew (inv(B) has changed)
            // the listed columns, ascending: place of this lane's column of slot c in the list
            int n = 0;
            unsigned long long mc[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                mc[c] = __ballot(((nbw & nzw) >> c) & 1u);
                n += __popcll(mc[c]);
            }
#ifdef SSQP_PHASE_PROFILE
            w1a[12] += n;
            w1a[13] += 1;
#endif
            if (n <= TCAP) {
                // through LDS: the owners put their listed columns (and x_k) side by side in list order -- every listed
                // column of a slot at once --, then lane r reads a column's entries from ONE address each (a broadcast read)
                // and forms row r of Y[:,k] x_k: a tenth of the v_readlane traffic of fetching them lane by lane
                int base = 0;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if (mc[c]) {  // uniform
                        const int at = base + __popcll(mc[c] & ((1ull << lane) - 1ull));
                        if ((mc[c] >> lane) & 1ull) {
#pragma unroll
                            for (int t = 0; t < MC; ++t) AL[at * MCP + t] = a[c][t];
                            AL[at * MCP + MC] = xL[lane + 64 * c];
                            ib2[at] = lane + 64 * c;
                        }
                        base += __popcll(mc[c]);
                    }
                }
                wave_sync();
                kkv = (lane < n) ? ib2[lane < n ? lane : 0] : 0x7fffffff;
                // lane (row r, group g) forms rows r of the columns j = g, g + RGRP, ...
                for (int j0 = 0; j0 < n; j0 += RGRP) {  // uniform trip count
                    const int j = j0 + rgrp;
                    const bool on = rlane && j < n;
                    const double *__restrict__ cj = AL + (on ? j : 0) * MCP;
                    double y = 0.0;
#pragma unroll
                    for (int t = 0; t < MC; ++t) y += ivr[t] * cj[t];
                    if (on) tl[j * MC + rrow] = y * cj[MC];
                }
                wave_sync();
                // ... and lane r adds its row in list order (one rounded add per term)
                for (int p0 = 0; p0 < n; p0 += 8) {
                    double tv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) tv[q] = tl[((p0 + q < n) ? p0 + q : 0) * MC + (lane < MC ? lane : 0)];
#pragma unroll
                    for (int q = 0; q < 8; ++q) a2 += (p0 + q < n) ? tv[q] : 0.0;   // (+0.0 on a sum that is never -0.0)
                }
                cached = true;
                cnt = n;
                ordv = lane;
                used = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
                wave_sync();
            } else {  // more columns at a nonzero bound than the cache holds: lane by lane, every pass
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    unsigned long long m = mc[c];
                    while (m) {  // uniform
                        const int l = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const double xk = xL[l + 64 * c];
                        double y = 0.0;
#pragma unroll
                        for (int t = 0; t < MC; ++t) y += ivr[t] * readlane_f64(a[c][t], l);
                        a2 += y * xk;
                    }
                }
                cached = false;
                cnt = 0;
            }
            rebuild = false;
            W1_STAMP(9);   // xb terms anew
        } else {  // the cached terms, re-added in ascending order of their columns, eight LDS reads per round trip
            for (int p0 = 0; p0 < cnt; p0 += 8) {
                double tv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int slot = __builtin_amdgcn_readlane(ordv, (p0 + q) & 63);
                    tv[q] = tl[slot * MC + (lane < MC ? lane : 0)];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) a2 += (p0 + q < cnt) ? tv[q] : 0.0;   // (+0.0 on a sum that is never -0.0)
            }
        }
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < MC; ++t) s += ivr[t] * readlane_f64(rhs, t);
        xb = s - a2;
        W1_STAMP(7);  // xb
    }
    W1_FLUSH();

    // ---- finish(): values of the basic variables; then initQP's mapping back (SSQP.jl:531-559)
    wave_sync();
    if (status >= 0 && lane < M0) xL[bas] = xb;
    wave_sync();
    int feasible = 1;
    if (status >= 0) {
        double art = 0.0;
        for (int kk = N0; kk < N1; ++kk) art += xL[kk];
        feasible = (art > tol) ? 0 : 1;
    }
    const bool mapBack = status >= 0 && feasible == 1;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = lane + 64 * c;
        const int sk = (int)((stw >> (2 * c)) & 3u);
        if (k < N) {
            double xk = xL[k];
            if (mapBack && ((upw >> c) & 1u)) xk = -xk;   // (statuses stay: SSQP.jl:552-557 is a no-op)
            x0[k] = xk;
            S[k] = sk;
        } else if (k < N0) {
            S[k] = mapBack ? ((sk == SSQP_IN) ? SSQP_OE : SSQP_EO) : sk;
        }
    }
    statusOut = status < 0 ? -1 : feasible;
    if (lane == 0) P.status[prob] = statusOut;
    wave_sync();
    return true;
}

}  // namespace p1w
}  // namespace ssqp
#endif
