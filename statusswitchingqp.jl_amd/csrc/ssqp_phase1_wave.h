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

// LDS of one wavefront, in doubles: lo, hi by variable id (64 NC each), the basis matrix by columns and the MC x MC
// transposes, MC scratch, then ints
template <int NC, int MC>
__host__ __device__ constexpr int lds_doubles() {
    return 2 * 64 * NC + 2 * MC * MC + 2 * MC + 2 * MC;  // (+ 2 MC doubles' worth of ints)
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
    const int lane = threadIdx.x & 63;
    int pv[MC];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < MC; ++k) {
        pv[k] = k;
        if (k < n && ok) {  // uniform
            double ck[MC];
#pragma unroll
            for (int i = k; i < MC; ++i) ck[i] = readlane_f64(col[i], k);
            int p = k;  // the FIRST largest |a(i, k)|, i = k .. n - 1 (the host's strict ">" scan)
            double best = fabs(ck[k]);
#pragma unroll
            for (int i = k + 1; i < MC; ++i)
                if (i < n) {
                    const double v = fabs(ck[i]);
                    if (v > best) best = v, p = i;
                }
            p = __builtin_amdgcn_readfirstlane(p);
            if (best == 0.0) {
                ok = false;
            } else {
                pv[k] = p;
                if (p != k) {  // rows k and p change places in every column (uniform p: one branch per candidate row)
#pragma unroll
                    for (int i = k + 1; i < MC; ++i)
                        if (i == p) {
                            const double t = col[k];
                            col[k] = col[i];
                            col[i] = t;
                            const double tc = ck[k];
                            ck[k] = ck[i];
                            ck[i] = tc;
                        }
                }
                const double r = 1.0 / ck[k];
                const double akj = col[k];
#pragma unroll
                for (int i = k + 1; i < MC; ++i)
                    if (i < n) {
                        const double li = ck[i] * r;                       // a(i, k) *= r
                        const double upd = col[i] - li * akj;              // a(i, j) -= a(i, k) * a(k, j)   (j > k)
                        col[i] = (lane == k) ? li : ((lane > k) ? upd : col[i]);
                    }
            }
        }
    }
    if (!ok) return false;
    // column `lane` of the inverse: L U x = P e_lane.  P e_c: the host applies the row swaps to e_c in order; the permuted
    // unit vector has its 1 where that sequence of swaps sends index c
    int pos = lane;
#pragma unroll
    for (int k = 0; k < MC; ++k)
        if (k < n && pv[k] != k) pos = (pos == k) ? pv[k] : ((pos == pv[k]) ? k : pos);
    double xc[MC];
#pragma unroll
    for (int i = 0; i < MC; ++i) xc[i] = (i == pos) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < MC; ++k) {  // forward: x[i] -= L(i, k) x[k]  for i > k
        if (k < n) {
#pragma unroll
            for (int i = k + 1; i < MC; ++i)
                if (i < n) {
                    const double l = readlane_f64(col[i], k);
                    xc[i] -= l * xc[k];
                }
        }
    }
#pragma unroll
    for (int kk = 0; kk < MC; ++kk) {  // backward: x[k] /= U(k, k), then x[i] -= U(i, k) x[k]  for i < k
        const int k = MC - 1 - kk;
        if (k < n) {
            const double ukk = readlane_f64(col[k], k);
            xc[k] /= ukk;
#pragma unroll
            for (int i = 0; i < MC; ++i)
                if (i < k) {
                    const double u = readlane_f64(col[i], k);
                    xc[i] -= u * xc[k];
                }
        }
    }
#pragma unroll
    for (int i = 0; i < MC; ++i) col[i] = xc[i];
    return true;
}

// One QP.  `lds`: lds_doubles<NC, MC>() doubles of this wavefront's LDS.  Returns false when the QP was not taken (it has
// free variables and now sits on P.fbList); otherwise x0, S and status are written.
template <int NC, int MC>
__device__ __forceinline__ bool solve_one(const Params &P, int prob, double *lds) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
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

    double *lo = lds, *hi = lo + 64 * NC;
    double *T = hi + 64 * NC;            // MC x MC: columns on their way between lanes
    double *Bc = T + MC * MC;            // MC x MC: column j of the basis matrix A1[:, basis[j]] at Bc[j * MC ..]
    double *sg = Bc + MC * MC;           // MC: signs of the artificial columns
    int *ib = reinterpret_cast<int *>(sg + 2 * MC);  // 2 MC ints

    // ---- the LP of initQP (SSQP.jl:484-526): columns [A; G | slack | artificials], this lane's columns in registers
    double a[NC][MC], sd[NC], xv[NC], nrm[NC];
    unsigned stw = 0, nbw = 0, upw = 0;   // per column slot: status (2 bits: IN 0, DN 1, UP 2), nonbasic, (-inf, u] variable
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
        xv[c] = lok;                          // x[k] = lo[k]  (every nonbasic starts at its lower bound)
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
    const double rhs = (M0 > 0) ? ((rr < M) ? b[rr < M ? rr : 0] : g[rr >= M ? rr - M : 0]) : 0.0;
    // start = sum over the columns k < N0 with lo != 0, ascending, of A1[:,k] * lo[k]   (SSQP.jl:511-526)
    double start = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = lane + 64 * c;
        unsigned long long m = __ballot(k < N0 && xv[c] != 0.0);
        while (m) {  // uniform
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const double lk = readlane_f64(xv[c], l);
            double ar = 0.0;
#pragma unroll
            for (int t = 0; t < MC; ++t)
                if (t < M0) {
                    const double v = readlane_f64(a[c][t], l);
                    ar = (lane == t) ? v : ar;
                }
            start += ar * lk;
        }
    }
    const double sgn = rhs >= start ? 1.0 : -1.0;
    if (lane < MC) sg[lane] = sgn;
    wave_sync();
    double ivr[MC];
#pragma unroll
    for (int t = 0; t < MC; ++t) {
        ivr[t] = (lane == t && lane < M0) ? sgn : 0.0;   // invB = diag(sgn)
        if (lane < MC) Bc[lane * MC + t] = ivr[t];       // column `lane` of the basis matrix: the artificial column
    }
    double xb = fabs(start - rhs);
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
        for (int r = 0; r < MC; ++r)
            if (r < M0) s += a[c][r] * a[c][r];
        nrm[c] = sqrt(s);
    }

    // Y = invB * A[:, nonbasic] is never stored (as in ssqp_phase1.hip): the pricing needs Y[:,k] . c[basis], formed
    // here for every column this lane owns, rows in the host's order; row r of invB and c[basis[r]] come out of lane r as
    // scalars
    auto refreshY = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < NC; ++c) sd[c] = 0.0;
        for (int r = 0; r < M0; ++r) {
            // c[basis[r]] is 1 for an artificial variable and 0 otherwise.  A row with c = 0 adds s * 0.0 = +-0.0 to a sum that
            // started at +0.0 and is never -0.0 (x + y = -0.0 needs x = y = -0.0): it changes no bit of it and is skipped --
            // more than half of the rows, on average (only a row sum that is Inf / NaN would have left a NaN behind)
            if (__builtin_amdgcn_readlane(bas, r) < N0) continue;
            const double cb = 1.0;
            double s[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) s[c] = 0.0;
#pragma unroll
            for (int t = 0; t < MC; ++t)
                if (t < M0) {
                    const double iv = readlane_f64(ivr[t], r);
#pragma unroll
                    for (int c = 0; c < NC; ++c) s[c] += iv * a[c][t];
                }
#pragma unroll
            for (int c = 0; c < NC; ++c) sd[c] += s[c] * cb;
        }
    };
    refreshY();

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
        const bool bland = loop > N1;
        // ---- price: signed reduced costs, the entering candidate (first maximum of h / ||A[:,k]||; Bland: first candidate)
        double best = -INF;
        int bidx = 0x7fffffff;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = lane + 64 * c;
            const double ck = k >= N0 ? 1.0 : 0.0;  // (= cost[k]: the Phase-1 objective is the sum of the artificials)
            double hv = ck - sd[c];
            if (((stw >> (2 * c)) & 3u) == (unsigned)SSQP_DN) hv = -hv;
            if (((nbw >> c) & 1u) && hv > tol) {
                const double v = bland ? 0.0 : hv / nrm[c];
                if (v > best) best = v, bidx = k;   // (ascending k inside the lane: the first maximum stays)
            }
        }
        {
            const KeyMin km = wave_keymin(KeyMin{-best, bidx});
            bidx = km.ord;
        }
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
                for (int t = 0; t < MC; ++t)
                    if (t < M0) ak[t] = readlane_f64(a[c][t], lk);
            }
        double p = 0.0;
#pragma unroll
        for (int t = 0; t < MC; ++t)
            if (t < M0) p += ivr[t] * ak[t];
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
        if (unbounded) {
            status = 3;
            break;
        }
        if (action < 0) {  // bound flip of the entering variable
            const unsigned ns = action == -1 ? (unsigned)SSQP_UP : (unsigned)SSQP_DN;
            if (lane == lk) stw = (stw & ~(3u << (2 * ck_))) | (ns << (2 * ck_));
            col_set<NC>(xv, k, action == -1 ? hiK : loK);
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
            for (int i = 0; i < MC; ++i)
                if (i < M0) rank += (__builtin_amdgcn_readlane(bas, i) < bas) ? 1 : 0;
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
            if (!invert_lu_regs<MC>(col, M0)) {  // lu() of the reference throws (Simplex.jl:590)
                status = -1;
                break;
            }
            // lane j holds column j of the inverse; row r goes to lane r
            if (lane < M0) {
#pragma unroll
                for (int t = 0; t < MC; ++t) T[t * MC + lane] = col[t];   // T[i][j] = inv(i, j)
            }
            wave_sync();
#pragma unroll
            for (int t = 0; t < MC; ++t) ivr[t] = (lane < M0 && t < M0) ? T[rr * MC + t] : 0.0;
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
                }
                col_set<NC>(xv, leaving, newx);
            }
            refreshY();
        }
        // ---- xb = invB * b - Y * x[nonbasic]: the nonbasic columns at a nonzero value, ascending, one rounded multiply and
        // one rounded add per term (Simplex.jl:599)
        double a2 = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            unsigned long long m = __ballot(((nbw >> c) & 1u) && xv[c] != 0.0);
            while (m) {  // uniform
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const double xk = readlane_f64(xv[c], l);
                double y = 0.0;
#pragma unroll
                for (int t = 0; t < MC; ++t)
                    if (t < M0) y += ivr[t] * readlane_f64(a[c][t], l);
                a2 += y * xk;
            }
        }
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < MC; ++t)
            if (t < M0) s += ivr[t] * readlane_f64(rhs, t);
        xb = s - a2;
    }

    // ---- finish(): values of the basic variables; then initQP's mapping back (SSQP.jl:531-559)
    if (status >= 0) {
        for (int j = 0; j < M0; ++j) col_set<NC>(xv, __builtin_amdgcn_readlane(bas, j), readlane_f64(xb, j));
    }
    int feasible = 1;
    if (status >= 0) {
        double art = 0.0;
        for (int kk = N0; kk < N1; ++kk) art += col_get<NC>(xv, kk);
        feasible = (art > tol) ? 0 : 1;
    }
    const bool mapBack = status >= 0 && feasible == 1;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = lane + 64 * c;
        const int sk = (int)((stw >> (2 * c)) & 3u);
        if (k < N) {
            double xk = xv[c];
            if (mapBack && ((upw >> c) & 1u)) xk = -xk;   // (statuses stay: SSQP.jl:552-557 is a no-op)
            x0[k] = xk;
            S[k] = sk;
        } else if (k < N0) {
            S[k] = mapBack ? ((sk == SSQP_IN) ? SSQP_OE : SSQP_EO) : sk;
        }
    }
    if (lane == 0) P.status[prob] = status < 0 ? -1 : feasible;
    wave_sync();
    return true;
}

}  // namespace p1w
}  // namespace ssqp
#endif
