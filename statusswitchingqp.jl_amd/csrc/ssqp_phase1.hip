// ssqp_phase1.hip -- gfx950: Phase-1 of solveQP(Q) for a BATCH of QPs on the GPU, one workgroup per QP: 256 threads and four
// workgroups per CU for M + J <= 12 rows, 512 threads (two wavefronts per SIMD) and one workgroup per CU for more rows -- the
// build whose inv(lu(B)), Y.c refresh and xb sum know that a simplex basis is mostly unit columns (see invert_lu, refreshY).
// (The headline shapes go through the one-wavefront-per-QP kernel, ssqp_phase1_wave.h; this file takes every other shape.)
//
// Replaces initQP (reference: src/SSQP.jl:461-560) and the bounded-variable simplex it calls, cDantzigLP
// (src/Simplex.jl:445-615), as the host C++ version in ssqp_host.cpp (phase1_one / BoundedSimplex) does -- and is
// BIT-IDENTICAL to it: the vertex (x0, S0) is what the active-set loop starts from, and the loop's pass count
// depends on it, so every decision (largest-distance Dantzig pricing with the switch to Bland's rule after N loops,
// first-minimum ratio test, bound flips, sorted basis, inv(lu(A[:,B])) with partial pivoting) and every rounding
// has to be the host's.  That fixes the arithmetic: each sum runs in the host's order with separately rounded
// multiply and add (this file is compiled with -ffp-contract=off), IEEE division and square root.  What the GPU adds is
// width: every column of the LP (N + J + n + M0 of them) belongs to one thread -- its entries of Y = invB*A[:,k],
// its reduced cost, its candidate ratio -- and the only sequential pieces are the ones the summation order forces
// (the ratio test over the M0 basic rows, the pivot search of the LU, the sums over the nonbasic columns that sit
// at a nonzero bound, taken from a compacted list).
//
// Workspace per QP in global memory (76 KB at N = 512, M0 = 11): the LP matrix A1 (M0 x N1, row-major) and the
// N1-vectors; the host's Y = invB * A[:, nonbasic] is never stored (see refreshY); LDS holds invB, the basis matrix
// being inverted and the M0-vectors.
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssqp_hip.h"
#include "ssqp_internal.h"
#include "ssqp_device.h"

namespace ssqp {
namespace p1 {

constexpr int NT1 = 256;   // threads per workgroup, few rows
template <bool BIG> constexpr int NTB = BIG ? 512 : NT1;   // many rows: two wavefronts per SIMD (what bounds that build is
                                                            // instruction issue: a lone wavefront issues once in ~4.5 cycles)
constexpr int XB_CHUNK = 32;   // columns of the xb sum whose products are formed together (staged in LDS)
constexpr int XB_CHUNK_BIG = 40;   // ... in the many-rows build: cfg5 lists 25-40 columns, one chunk instead of two
constexpr double INF = __builtin_huge_val();

// ---- diagnostic build only (-DSSQP_PHASE_PROFILE): cycles per phase of the kernel, thread 0 of every workgroup ----
#ifdef SSQP_PHASE_PROFILE
static __device__ unsigned long long g_p1phase[24];
#define P1_DECL unsigned long long p1t = __builtin_amdgcn_s_memtime(); unsigned long long p1a[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define P1_STAMP(slot)                                                   \
    do {                                                                 \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();      \
        p1a[slot] += t_ - p1t;                                           \
        p1t = t_;                                                        \
    } while (0)
#define P1_COUNT(slot) p1a[slot] += 1
#define P1_FLUSH()                                                                                                     \
    do {                                                                                                               \
        if (tid == 0)                                                                                                  \
            for (int k_ = 0; k_ < 16; ++k_)                                                                            \
                (void)__hip_atomic_fetch_add(&g_p1phase[k_], p1a[k_], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     \
    } while (0)
#else
#define P1_DECL do { } while (0)
#define P1_STAMP(slot) do { } while (0)
#define P1_COUNT(slot) do { } while (0)
#define P1_FLUSH() do { } while (0)
#endif

// block-wide (value, index) maximum with the FIRST maximum winning (smallest index on ties); all threads get it
template <int NT>
__device__ __forceinline__ void block_first_max(double &v, int &idx, double *rv, int *ri) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {   // inside the wavefront: DPP steps (the first maximum of v = the first minimum of -v)
        const KeyMin km = wave_keymin(KeyMin{-v, idx});
        v = -km.v;
        idx = km.ord;
    }
    if (lane == 0) {
        rv[wave] = v;
        ri[wave] = idx;
    }
    __syncthreads();
    v = rv[0];
    idx = ri[0];
    for (int w = 1; w < NT / 64; ++w) {
        const bool take = (rv[w] > v) || (rv[w] == v && ri[w] < idx);
        v = take ? rv[w] : v;
        idx = take ? ri[w] : idx;
    }
    __syncthreads();
}

// ascending list of the columns k < n with pred(k): one wavefront scans in chunks of 64 (ballot + popcount keeps
// the order); returns the count to every thread through *cnt (LDS)
template <class Pred>
__device__ __forceinline__ int compact_columns(int tid, int n, int *list, int *cnt, Pred pred) {
    if (tid < 64) {
        const int lane = tid;
        int base = 0;
        for (int cb = 0; cb < n; cb += 512) {  // (eight chunks' flags are read together: their reads do not wait for the list's writes)
            bool f[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = cb + 64 * q + lane;
                f[q] = (k < n) && pred(k);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned long long m = __ballot(f[q]);
                if (f[q]) list[base + __popcll(m & ((1ull << lane) - 1ull))] = cb + 64 * q + lane;
                base += __popcll(m);
            }
        }
        if (lane == 0) *cnt = base;
    }
    __syncthreads();
    return *cnt;
}

// The same list by the whole workgroup (NT threads): every thread evaluates eight columns' flags at once -- ONE memory round trip
// per 8 NT columns instead of one per 512 -- the wavefronts' ballot counts go through LDS (wcnt: 8 NT / 64 ints), every thread
// forms its own bases from them, ascending k = (round, wavefront, lane).  Returns the count to every thread.
template <int NT, class Pred>
__device__ __forceinline__ int compact_columns_wg(int tid, int n, int *list, int *wcnt, Pred pred) {
    constexpr int NW = NT / 64, QR = 8;
    const int lane = tid & 63, wave = tid >> 6;
    int total = 0;
    for (int k0 = 0; k0 < n; k0 += QR * NT) {
        bool f[QR];
        unsigned long long m[QR];
#pragma unroll
        for (int q = 0; q < QR; ++q) {
            const int k = k0 + q * NT + tid;
            f[q] = (k < n) && pred(k);
        }
#pragma unroll
        for (int q = 0; q < QR; ++q) {
            m[q] = __ballot(f[q]);
            if (lane == 0) wcnt[q * NW + wave] = __popcll(m[q]);
        }
        __syncthreads();
        int run = total, mine[QR];
#pragma unroll
        for (int q = 0; q < QR; ++q)
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                if (w == wave) mine[q] = run;
                run += wcnt[q * NW + w];
            }
#pragma unroll
        for (int q = 0; q < QR; ++q)
            if (f[q]) list[mine[q] + __popcll(m[q] & ((1ull << lane) - 1ull))] = k0 + q * NT + tid;
        total = run;
        __syncthreads();
    }
    return total;
}

__device__ __forceinline__ void wave_order() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// inv(lu(a)) in place for the n x n column-major LDS matrix a (n <= 128), partial pivoting, the host's operation order per
// element (ssqp_host.cpp invert_lu).  Scratch: x (n x (n + 1) doubles), vec (2 n doubles), piv (2 n ints).  Returns false
// when a pivot is exactly 0.
//
// A simplex basis is mostly UNIT columns (slacks, artificials): cfg5's 72 x 72 bases hold 1 to 11 dense columns, their
// inverses 3 % to 30 % nonzeros.  The host's loops run over every element; here every step first lists what is NOT an exact
// zero, and an update a -= l * u with l or u exactly 0 is left out.  That changes no bit of the inverse's nonzero entries
// and no decision taken on it: the only thing an update by +-0.0 can do is turn a -0.0 into +0.0, the trailing matrix is
// only ever subtracted from (a difference is never -0.0 unless its first operand was), and a zero of either sign gives the
// same pivot search (|.|), the same products (+-0.0, added to sums that are never -0.0) and the same "pivot is exactly 0"
// verdict.  Likewise a column entry of the inverse that is 0 before its division stays +0.0 where the host has 0 / U = +-0.0.
//   * elimination step k: wavefront 0 looks at column k.  ONE candidate that is not exactly zero: a unit column -- that entry
//     is the pivot, the L column is zero, the step is an exchange of two rows and nothing else; a run of such steps is walked
//     by wavefront 0 alone with the rows' logical positions in registers (see the loop).  More: the pivot by a wavefront
//     reduction (first maximum: value, then smallest row among the ties), the nonzero rows of the L column listed by ballot;
//     after ONE barrier the rows k and p change places in every column and column k is scaled (distinct threads, distinct
//     elements, 1 / pivot by every scaling thread); after a second the trailing update runs over (listed rows) x (columns with
//     a nonzero entry in the U row), and a third barrier ends the step;
//   * columns of the inverse: three (six) lanes of one wavefront per column, no barrier between steps; a
//     forward step whose L column is zero is not walked at all (a bit mask in scalar registers), neither is a step at which
//     none of the wavefront's columns has a nonzero entry; the backward pass walks only the steps with a nonempty U column, the
//     divisions of the other entries wait until the end (nobody reads them before).
template <int NT>
__device__ __forceinline__ bool invert_lu(int tid, double *a, double *x, double *vec, int *piv, int n, int *flag) {
    const int lane = tid & 63;
    // (vec: 4 n doubles = 8 n ints)
    int *Lnz = reinterpret_cast<int *>(vec), *hasU = Lnz + 2 * n, *physOf = Lnz + 3 * n, *mv = Lnz + 4 * n;
    int *ctrl = Lnz + 5 * n;   // [0] the step the workgroup takes up (n: none is left), [1] where the run started, [2] rows to move
    int *stepInfo = piv + n;                                                     // nL of step k (0: an exchange only)
#ifdef SSQP_PHASE_PROFILE
    const unsigned long long luT0 = __builtin_amdgcn_s_memtime();
#endif
    if (tid < 64) {
        for (int j = lane; j < n; j += 64) hasU[j] = 0;
        wave_order();
    }
    for (int k = 0;;) {
#ifdef SSQP_PHASE_PROFILE
        const unsigned long long eT0 = __builtin_amdgcn_s_memtime();
#endif
        if (tid < 64) {
            // Wavefront 0 walks the steps from k on.  A step whose L column is zero updates nothing: its whole effect is the
            // exchange of two rows.  During a run of such steps nothing in the matrix changes, so the exchange is only NOTED --
            // every lane keeps the logical position of its two physical rows in registers, the pivot search orders its
            // candidates by logical row -- and a step costs one column read and one wavefront reduction, no LDS write in the
            // dependent chain and no barrier.  When a step with a nonzero L column turns up (or the last step is through)
            // the workgroup moves the rows that changed places, once, and marks the U rows of the run.
            const int r0 = lane, r1 = lane + 64;
            int lg0 = r0, lg1 = r1;
            bool anySwap = false, ok = true, dense = false;
            int kk = k;
            double c0 = r0 < n ? a[(size_t)kk * n + r0] : 0.0, c1 = r1 < n ? a[(size_t)kk * n + r1] : 0.0;
            for (;;) {
                const bool v0 = r0 < n && lg0 >= kk, v1 = r1 < n && lg1 >= kk;
                // (the next column, in case this step turns out to be an exchange only: nothing is written until then)
                const int kn = kk + 1 < n ? kk + 1 : kk;
                const double n0 = r0 < n ? a[(size_t)kn * n + r0] : 0.0, n1 = r1 < n ? a[(size_t)kn * n + r1] : 0.0;
                // how many candidates are not exactly zero?  One: it is the pivot and the L column is zero -- the step of a unit
                // column, most steps -- and no reduction is needed to find it.  None: the pivot is exactly 0.
                const unsigned long long z0 = __ballot(v0 && c0 != 0.0), z1 = __ballot(v1 && c1 != 0.0);
                const int nnz = __popcll(z0) + __popcll(z1);
                if (nnz == 0) {
                    ok = false;
                    break;
                }
                if (nnz == 1) {
                    const int p = z0 ? __builtin_amdgcn_readlane(lg0, __builtin_ctzll(z0)) : __builtin_amdgcn_readlane(lg1, __builtin_ctzll(z1));
                    if (p != kk) {
                        lg0 = lg0 == kk ? p : (lg0 == p ? kk : lg0);
                        lg1 = lg1 == kk ? p : (lg1 == p ? kk : lg1);
                        anySwap = true;
                    }
                    if (lane == 0) {
                        piv[kk] = p;
                        stepInfo[kk] = 0;
                    }
                    if (++kk == n) break;
                    c0 = n0;
                    c1 = n1;
                    continue;
                }
                // the FIRST largest |a(i, kk)| over the logical rows i = kk .. n - 1 (the host's strict ">" scan)
                const KeyMin cur = keymin(KeyMin{v0 ? -fabs(c0) : 1.0, v0 ? lg0 : 0x7fffffff}, KeyMin{v1 ? -fabs(c1) : 1.0, v1 ? lg1 : 0x7fffffff});
                const KeyMin km = wave_keymin(cur);
                ok = !(km.v == 0.0 || km.ord >= n);   // (no order: nothing but NaN)
                if (!ok) break;
                const int p = km.ord;
                const bool l0 = v0 && lg0 != p && c0 != 0.0, l1 = v1 && lg1 != p && c1 != 0.0;   // nonzeros of the L column: at least one
                const unsigned long long m0 = __ballot(l0), m1 = __ballot(l1);
                // a step with a nonzero L column: its lists in LOGICAL rows (what is physical once the rows have been moved),
                // at the positions after this step's own exchange of kk and p
                dense = true;
                const unsigned long long below = (1ull << lane) - 1ull;
                if (l0) Lnz[__popcll(m0 & below)] = lg0 == kk ? p : lg0;
                if (l1) Lnz[__popcll(m0) + __popcll(m1 & below)] = lg1 == kk ? p : lg1;
                const int nL = __popcll(m0) + __popcll(m1);
                // (the U row's nonzero columns are found by the update itself and 1 / pivot by the threads that scale the column:
                //  both would be a dependent LDS round trip and a division in THIS wavefront's chain, which every step waits for)
                if (lane == 0) {
                    piv[kk] = p;
                    stepInfo[kk] = nL;
                }
                break;
            }
            int nMv = 0;
            if (ok && anySwap) {  // where every logical row lies, and the list of those that have to move
                if (r0 < n) physOf[lg0] = r0;
                if (r1 < n) physOf[lg1] = r1;
                const bool f0 = r0 < n && lg0 != r0, f1 = r1 < n && lg1 != r1;
                const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (f0) mv[__popcll(m0 & below)] = lg0;
                if (f1) mv[__popcll(m0) + __popcll(m1 & below)] = lg1;
                nMv = __popcll(m0) + __popcll(m1);
            }
            if (lane == 0) {
                *flag = ok ? 1 : 0;
                ctrl[0] = kk;
                ctrl[1] = k;
                ctrl[2] = nMv;
                ctrl[3] = dense ? 1 : 0;
            }
        }
        __syncthreads();
#ifdef SSQP_PHASE_PROFILE
        const unsigned long long eT1 = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            (void)__hip_atomic_fetch_add(&g_p1phase[17], eT1 - eT0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_add(&g_p1phase[16], (unsigned long long)(ctrl[3] != 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_add(&g_p1phase[19], (unsigned long long)(ctrl[2] != 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
        if (!*flag) return false;
        const int k0 = k, nMv = ctrl[2];
        const bool dense = ctrl[3] != 0;
        k = ctrl[0];
        // the U rows of the run's steps k0 .. k - 1 (logical row i lies at physOf[i] while rows wait to be moved): a column with
        // an entry in one of them has a nonempty U column
        for (int e = tid; e < (k - k0) * n; e += NT) {
            const int di = e / n, j = e - di * n, i = k0 + di;
            if (j > i && a[(size_t)j * n + (nMv ? physOf[i] : i)] != 0.0) hasU[j] = 1;
        }
        if (nMv) {  // the rows that changed places, through the scratch (read everything, barrier, write)
            for (int e = tid; e < nMv * n; e += NT) {
                const int m = e / n, j = e - m * n;
                x[e] = a[(size_t)j * n + physOf[mv[m]]];
            }
            __syncthreads();
            for (int e = tid; e < nMv * n; e += NT) {
                const int m = e / n, j = e - m * n;
                a[(size_t)j * n + mv[m]] = x[e];
            }
        }
        if (!dense) break;   // (the last step is through; the barrier before the columns of the inverse orders the writes above)
        if (nMv) __syncthreads();   // (rows were written: the exchange below reads them; the marks alone need no barrier)
        const int p = piv[k], info = stepInfo[k];
        // rows k and p change places in every column but k (thread j < 128 takes column j); column k: a(i, k) = a(i', k) / pivot
        // for i > k with i' the row the swap brings to i, and the pivot itself moves to (k, k) (thread 128 + i - k - 1)
        if (tid < 128) {
            if (tid < n && tid != k && p != k) {
                double *colj = a + (size_t)tid * n;
                const double t = colj[k];
                colj[k] = colj[p];
                colj[p] = t;
            }
        } else {
            const int i = k + 1 + (tid - 128);
            if (i < n) {
                double *colk = a + (size_t)k * n;
                const double src = colk[i == p ? k : i], pv = colk[p];
                const double r = 1.0 / pv;       // (every thread: cheaper than one division in wavefront 0's chain and a broadcast)
                colk[i] = src * r;               // a(i, k) *= 1 / a(k, k)
                if (i == p) colk[k] = pv;
            }
        }
        __syncthreads();
        const int nL = info & 255;
        {
            // a(i, j) -= a(i, k) * a(k, j) over the listed rows and the columns with a nonzero entry in the U row: thread
            // (cj = tid >> 4, ri = tid & 15) takes the listed rows ri + 16 m of the columns k + 1 + cj + (NT / 16) g, five rows per
            // LDS round trip; a column with an entry in the U row has a nonempty U column
            const int cj = tid >> 4, ri = tid & 15;
            const double *colk = a + (size_t)k * n;
            for (int j = k + 1 + cj; j < n; j += NT / 16) {
                double *colj = a + (size_t)j * n;
                const double uj = colj[k];
                if (uj == 0.0) continue;
                if (ri == 0) hasU[j] = 1;
                for (int i0 = ri; i0 < nL; i0 += 16 * 5) {
                    int ix[5];
                    double ov[5], lv[5];
#pragma unroll
                    for (int m = 0; m < 5; ++m) ix[m] = Lnz[i0 + 16 * m < nL ? i0 + 16 * m : i0];
#pragma unroll
                    for (int m = 0; m < 5; ++m) {
                        ov[m] = colj[ix[m]];
                        lv[m] = colk[ix[m]];
                    }
#pragma unroll
                    for (int m = 0; m < 5; ++m)
                        if (i0 + 16 * m < nL) colj[ix[m]] = ov[m] - lv[m] * uj;
                }
            }
            __syncthreads();
        }
#ifdef SSQP_PHASE_PROFILE
        if (tid == 0) (void)__hip_atomic_fetch_add(&g_p1phase[18], __builtin_amdgcn_s_memtime() - eT1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        if (++k >= n) break;
    }
    __syncthreads();
    // columns of the inverse: L U x_c = P e_c, THREE lanes of one wavefront per column (lane q takes the rows i = q mod 3), 21
    // columns per wavefront: between two steps the three only need the wavefront's own LDS ordering, no barrier.  P e_c: the
    // host applies the row swaps to e_c in order; the permuted unit vector has its 1 where that sequence of swaps sends c
    const int xs = n + 1;
#ifdef SSQP_PHASE_PROFILE
    const unsigned long long luT1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) (void)__hip_atomic_fetch_add(&g_p1phase[10], luT1 - luT0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    auto columns = [&](auto lpcTag) {
        // (LPC lanes per column, lane q of them takes the rows i = q mod LPC; 64 / LPC columns per wavefront)
        constexpr int LPC = decltype(lpcTag)::value, CPW = 64 / LPC;
        const int wv = tid >> 6;
        const int c = wv * CPW + lane / LPC, q = lane % LPC;
        const bool mine = lane < CPW * LPC && c < n;
        // the steps with a nonzero L column / a nonempty U column, as bit masks (steps 0 .. 63, 64 .. 127)
        unsigned long long fwd[2], upd[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kk = 64 * h + lane;
            fwd[h] = __ballot(kk < n && (stepInfo[kk < n ? kk : 0] & 255) != 0);
            upd[h] = __ballot(kk < n && hasU[kk < n ? kk : 0] != 0);
        }
        double *xc = x + (size_t)(mine ? c : 0) * xs;
        if (mine) {
            int pos = c;
            for (int k = 0; k < n; ++k) {
                const int pk = piv[k];
                if (pk != k) pos = (pos == k) ? pk : ((pos == pk) ? k : pos);
            }
            for (int i = q; i < n; i += LPC) xc[i] = (i == pos) ? 1.0 : 0.0;
        }
        wave_order();
        // x[i] -= f[i] * t for this lane's rows of lo .. hi - 1, eight per LDS round trip (the compiler cannot tell that the
        // stores to x do not alias the factors and would wait for every element's own read - modify - write)
        auto axpy = [&](const double *f, int lo, int hi, double t) {
            int first = lo + ((q - lo) % LPC + LPC) % LPC;   // the first row >= lo with i = q mod LPC
            for (int i0 = first; i0 < hi; i0 += 8 * LPC) {
                double fv[8], xv[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int i = i0 + LPC * m < hi ? i0 + LPC * m : i0;
                    fv[m] = f[i];
                    xv[m] = xc[i];
                }
#pragma unroll
                for (int m = 0; m < 8; ++m)
                    if (i0 + LPC * m < hi) xc[i0 + LPC * m] = xv[m] - fv[m] * t;
            }
        };
        // forward: x[i] -= L(i, k) x[k]  for i > k  (the host's "t != 0" guard only skips zeros; so does the mask)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned long long m = fwd[h];
            while (m) {
                const int k = 64 * h + __builtin_ctzll(m);
                m &= m - 1ull;
                const double t = xc[k];
                if (__ballot(mine && t != 0.0) == 0ull) continue;   // (none of this wavefront's columns has an entry here)
                if (mine) axpy(a + (size_t)k * n, k + 1, n, t);
                wave_order();
            }
        }
        // backward: x[k] /= U(k, k), then x[i] -= U(i, k) x[k]  for i < k -- over the steps with a nonempty U column, last
        // first.  An entry whose own U column is empty updates nobody: its division waits until every step that updates IT is
        // through and is then one of many independent ones (same operands as at the host's step k, so the same bits)
#pragma unroll
        for (int h = 1; h >= 0; --h) {
            unsigned long long m = upd[h];
            while (m) {
                const int bit = 63 - __builtin_clzll(m);
                const int k = 64 * h + bit;
                m &= ~(1ull << bit);
                const double xk = xc[k];
                if (__ballot(mine && xk != 0.0) == 0ull) continue;
                const double t = xk / a[(size_t)k * n + k];
                wave_order();   // (the three lanes have read x[k])
                if (mine) {
                    if (k % LPC == q) xc[k] = t;
                    axpy(a + (size_t)k * n, 0, k, t);
                }
                wave_order();
            }
        }
        if (mine) {
            for (int i0 = q; i0 < n; i0 += 8 * LPC) {
                double xv[8], dv[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int i = i0 + LPC * m < n ? i0 + LPC * m : i0;
                    xv[m] = xc[i];
                    dv[m] = a[(size_t)i * n + i];
                }
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int i = i0 + LPC * m;
                    if (i < n && !((upd[i >> 6] >> (i & 63)) & 1ull) && xv[m] != 0.0) xc[i] = xv[m] / dv[m];
                }
            }
        }
    };
    // (three lanes per column serve 21 columns per wavefront: 84 with four wavefronts, 168 with eight; with eight wavefronts and
    //  n <= 80, six lanes per column halve a step's rows per lane)
    if (NT >= 512 && n <= 80) columns(std::integral_constant<int, 6>{});
    else columns(std::integral_constant<int, 3>{});
    __syncthreads();
#ifdef SSQP_PHASE_PROFILE
    if (tid == 0) (void)__hip_atomic_fetch_add(&g_p1phase[11], __builtin_amdgcn_s_memtime() - luT1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    for (int e = tid; e < n * n; e += NT) {
        const int cc = e / n, i = e - cc * n;
        a[e] = x[(size_t)cc * xs + i];
    }
    __syncthreads();
    return true;
}

// The same inversion for n <= LUC by ONE wavefront with the matrix in REGISTERS: lane j holds column j.  As a workgroup
// every one of the ~3 n dependent steps ends in barriers (some ninety per basis change at n = 11) and an element is a
// round trip to LDS; here a step of the elimination is: lane k finds the pivot in its own registers and publishes it,
// every lane swaps its two entries, lane k scales and publishes its column, every lane right of it updates its own --
// two LDS broadcasts per step, no barrier (a wavefront's LDS operations execute in order; wave_order keeps the compiler
// from reordering them).  The columns of the inverse then take one lane each: the forward and backward substitution of
// column c touch only x_c, so lane c runs both in its registers, reading the factors from LDS.  Operations per element
// and their order exactly as above.  Called by the lanes of the first wavefront; returns false (to all of them) when a
// pivot is exactly 0.
constexpr int LUC = 12;
__device__ __forceinline__ bool invert_lu_cols(double *a, double *lbuf, int *piv, int n) {
    const int j = threadIdx.x & 63;
    const bool mine = j < n;
    double col[LUC];
#pragma unroll
    for (int i = 0; i < LUC; ++i) col[i] = a[(mine ? j : 0) * n + (i < n ? i : 0)];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < LUC; ++k) {
        if (k < n && ok) {  // uniform
            if (j == k) {  // the FIRST largest |a(i, k)|, i = k .. n - 1 (the host's strict ">" scan)
                int p = k;
                double best = fabs(col[k]);
#pragma unroll
                for (int i = k + 1; i < LUC; ++i)
                    if (i < n) {
                        const double v = fabs(col[i]);
                        if (v > best) best = v, p = i;
                    }
                piv[k] = (best == 0.0) ? -1 : p;
            }
            wave_order();
            const int p = piv[k];
            if (p < 0) {
                ok = false;
            } else {
                if (p != k) {  // rows k and p change places in every column (p > k)
                    double vp = col[k];
#pragma unroll
                    for (int i = k + 1; i < LUC; ++i) vp = (i == p) ? col[i] : vp;
                    const double vk = col[k];
                    col[k] = vp;
#pragma unroll
                    for (int i = k + 1; i < LUC; ++i) col[i] = (i == p) ? vk : col[i];
                }
                if (j == k) {
                    const double r = 1.0 / col[k];
#pragma unroll
                    for (int i = k + 1; i < LUC; ++i)
                        if (i < n) {
                            col[i] *= r;
                            lbuf[i] = col[i];
                        }
                }
                wave_order();
                double l[LUC];
#pragma unroll
                for (int i = k + 1; i < LUC; ++i) l[i] = lbuf[i < n ? i : k + 1 < n ? k + 1 : 0];  // (read together)
                if (mine && j > k) {
                    const double akj = col[k];
#pragma unroll
                    for (int i = k + 1; i < LUC; ++i)
                        if (i < n) col[i] -= l[i] * akj;
                }
                wave_order();
            }
        }
    }
    if (!ok) return false;
    if (mine) {
#pragma unroll
        for (int i = 0; i < LUC; ++i)
            if (i < n) a[j * n + i] = col[i];
    }
    wave_order();
    // column j of the inverse: L U x = P e_j
    int pv[LUC];
#pragma unroll
    for (int k = 0; k < LUC; ++k) pv[k] = piv[k < n ? k : 0];
    int pos = j;
#pragma unroll
    for (int k = 0; k < LUC; ++k)
        if (k < n && pv[k] != k) pos = (pos == k) ? pv[k] : ((pos == pv[k]) ? k : pos);
    double xc[LUC];
#pragma unroll
    for (int i = 0; i < LUC; ++i) xc[i] = (i == pos) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < LUC; ++k) {  // forward: x[i] -= L(i, k) x[k]  for i > k
        if (k < n) {
            double l[LUC];
#pragma unroll
            for (int i = k + 1; i < LUC; ++i) l[i] = a[k * n + (i < n ? i : k)];
#pragma unroll
            for (int i = k + 1; i < LUC; ++i)
                if (i < n) xc[i] -= l[i] * xc[k];
        }
    }
#pragma unroll
    for (int kk = 0; kk < LUC; ++kk) {  // backward: x[k] /= U(k, k), then x[i] -= U(i, k) x[k]  for i < k
        const int k = LUC - 1 - kk;
        if (k < n) {
            double u[LUC];
#pragma unroll
            for (int i = 0; i < LUC; ++i)
                if (i <= k) u[i] = a[k * n + i];
            xc[k] /= u[k];
#pragma unroll
            for (int i = 0; i < LUC; ++i)
                if (i < k) xc[i] -= u[i] * xc[k];
        }
    }
    wave_order();  // (every lane has read the factors)
    if (mine) {
#pragma unroll
        for (int i = 0; i < LUC; ++i)
            if (i < n) a[j * n + i] = xc[i];
    }
    wave_order();
    return true;
}

struct P1Params {
    int nprob, N, M, J;
    const double *A, *G, *b, *g, *d, *u;   // per problem, back to back (A: M x N, G: J x N, column-major)
    double tol;
    double *x0;
    int32_t *S;
    int32_t *status;
    double *ws;        // nprob workspaces of wsStride doubles
    size_t wsStride;
    int *wsInt;        // nprob integer workspaces of wsIntStride ints
    size_t wsIntStride;
    int ldsVec;        // 1: the N1-vectors every pass reads (S1, nonbasic, x, colnorm, sdot) live in LDS (they fit)
    // the QPs the one-wavefront-per-QP kernel (ssqp_phase1_wave.hip) did not take: problem ids and their count (null: all)
    const unsigned int *listCount;
    const int *list;
    // single-launch solveQP(Q) (ssqp_solve_full_batch_dev_f64): a QP of the list goes on into the loop as a hand-over at pass
    // 0 -- z = x0, fbIter = 0, zeroed statistics, its id appended to (hoCount, hoList) -- when its vertex is feasible, and
    // ends here with (x0, S, status) otherwise (SSQP.jl:229-232).  hoList null: plain Phase-1.
    unsigned int *hoCount;
    int *hoList;
    long long *hoIter;
    double *z;
    int64_t *status64;
    int32_t *detail;
    ssqp_stats *stats;
};

// BIG: the build for many rows (M0 > 12), one workgroup per CU: twice the registers, spent on wider tiles of the Y . c refresh
template <bool BIG>
__device__ __forceinline__ void phase1_one_wg(const P1Params &P, const int prob, unsigned char *smem) {
    constexpr int NT1 = NTB<BIG>;   // (this build's threads per workgroup)
    const int tid = threadIdx.x;
    P1_DECL;
    const int N = P.N, M = P.M, J = P.J, M0 = M + J;
    const double *A = P.A + (size_t)prob * M * N;
    const double *G = P.G + (size_t)prob * J * N;
    const double *b = P.b + (size_t)prob * M;
    const double *g = P.g + (size_t)prob * J;
    const double *d = P.d + (size_t)prob * N;
    const double *u = P.u + (size_t)prob * N;
    const double tol = P.tol;
    double *x0 = P.x0 + (size_t)prob * N;
    int32_t *S = P.S + (size_t)prob * (N + J);

    // ---- LDS: invB, Bm (M0 x M0 each), rhs, xb, pvec, start (M0 each), small integers
    double *invB = reinterpret_cast<double *>(smem);
    double *Bm = invB + (size_t)M0 * M0;
    double *rhs = Bm + (size_t)M0 * (M0 + (BIG ? 8 : 1));   // (Bm: M0 x (M0 + 1), the padded columns of the inverse while they are formed;
                                                             //  many rows: M0 x (M0 + 8), the refresh packs rows of inv(B) there)
    double *xb = rhs + M0;
    double *pv = xb + M0;
    double *acc = pv + M0;
    double *blo = acc + M0, *bhi = blo + M0;       // bounds of the basic variables by row
    double *redv = bhi + M0;                       // 4 (many rows: 12 -- one per wavefront for the block maximum, [8]: the leaving bound)
    double *terms = redv + (BIG ? 12 : 4);                      // XB_CHUNK x M0: products Y[r, k] x[k] of the xb sum, a chunk of columns at a time
    int *basis = reinterpret_cast<int *>(terms + (size_t)(BIG ? XB_CHUNK_BIG : XB_CHUNK) * M0);  // M0
    int *piv = basis + M0;                         // 2 M0 (row swaps of the LU, then the positions of the permuted unit vectors)
    int *redi = piv + 2 * M0;                      // 4 (many rows: 8)
    int *misc = redi + (BIG ? 8 : 4);                          // [0] count, [1] flag, [2] action, [3] leaveStatus, [4] n free, [5] n upperOnly

    // ---- free / upper-only variables (SSQP.jl:484-509), ascending lists
    int *iw = P.wsInt + (size_t)prob * P.wsIntStride;
    int *freeVars = iw, *upperOnly = iw + N;
    const int nfree = compact_columns(tid, N, freeVars, &misc[4], [&](int k) { return u[k] == INF && d[k] == -INF; });
    const int nup = compact_columns(tid, N, upperOnly, &misc[5], [&](int k) { return !(u[k] == INF && d[k] == -INF) && d[k] == -INF; });
    const int N0 = N + J + nfree, N1 = N0 + M0;
    int32_t *S1 = iw + 2 * N;
    int *nonbasic = S1 + N1;
    int *list = nonbasic + N1;
    double *wd = P.ws + (size_t)prob * P.wsStride;
    double *A1 = wd;
    double *lo = A1 + (size_t)M0 * N1;
    // M0: sign of the artificial column of every row (behind the integers, 8-byte aligned)
    double *sgnArt = reinterpret_cast<double *>((reinterpret_cast<size_t>(misc + 8) + 7) & ~(size_t)7);
    double *hi = lo + N1, *cost = hi + N1, *x = cost + N1, *colnorm = x + N1, *range = colnorm + N1;
    double *sdot = range + N1;   // per nonbasic column: Y[:,k] . c[basis] of the current basis (formed with Y, read by the pricing)
    if (P.ldsVec) {
        // the vectors every simplex pass sweeps (pricing, column lists) in LDS instead of the global workspace: a pass is
        // then free of dependent global round trips apart from the entering column
        const int N1x = 2 * N + J + M0;  // (largest N1: every variable free)
        double *dv = sgnArt + M0;
        x = dv;
        colnorm = dv + N1x;
        sdot = dv + 2 * (size_t)N1x;
        S1 = reinterpret_cast<int32_t *>(dv + 3 * (size_t)N1x);
        nonbasic = S1 + N1x;
    }

    // ---- the LP of initQP: A1 = [A; G | slack | -free copies | artificials]
    for (size_t e = tid; e < (size_t)M0 * N1; e += NT1) A1[e] = 0.0;
    for (int k = tid; k < N1; k += NT1) {
        lo[k] = 0.0;
        hi[k] = INF;
        cost[k] = 0.0;
        S1[k] = SSQP_DN;
    }
    __syncthreads();
    for (int k = tid; k < N; k += NT1) {
        for (int r = 0; r < M; ++r) A1[(size_t)(r) * N1 + k] = A[(size_t)k * M + r];
        for (int r = 0; r < J; ++r) A1[(size_t)(M + r) * N1 + k] = G[(size_t)k * J + r];
        lo[k] = d[k];
        hi[k] = u[k];
    }
    for (int j = tid; j < J; j += NT1) A1[(size_t)(M + j) * N1 + (N + j)] = 1.0;
    __syncthreads();
    for (int t = tid; t < nfree; t += NT1) {
        const int k = freeVars[t];
        for (int r = 0; r < M0; ++r) A1[(size_t)(r) * N1 + (N + J + t)] = -A1[(size_t)(r) * N1 + k];
        lo[k] = 0.0;
    }
    for (int t = tid; t < nup; t += NT1) {
        const int k = upperOnly[t];
        lo[k] = -hi[k];
        hi[k] = INF;
        for (int r = 0; r < M0; ++r) A1[(size_t)(r) * N1 + k] = -A1[(size_t)(r) * N1 + k];
    }
    for (int r = tid; r < M0; r += NT1) rhs[r] = (r < M) ? b[r] : g[r - M];
    __syncthreads();
    // start = sum over the columns with lo != 0, ascending, of A1[:,k] * lo[k]
    {
        const int cnt = compact_columns(tid, N0, list, &misc[0], [&](int k) { return lo[k] != 0.0; });
        for (int r = tid; r < M0; r += NT1) {
            double s = 0.0;
            for (int t = 0; t < cnt; ++t) {
                const int k = list[t];
                s += A1[(size_t)(r) * N1 + k] * lo[k];
            }
            acc[r] = s;
        }
        __syncthreads();
    }
    for (int e = tid; e < M0 * M0; e += NT1) invB[e] = 0.0;
    __syncthreads();
    for (int j = tid; j < M0; j += NT1) {
        const double sgn = rhs[j] >= acc[j] ? 1.0 : -1.0;
        sgnArt[j] = sgn;
        invB[(size_t)j * M0 + j] = sgn;
        A1[(size_t)(j) * N1 + (N0 + j)] = sgn;
        xb[j] = fabs(acc[j] - rhs[j]);
        basis[j] = N0 + j;
        S1[N0 + j] = SSQP_IN;
        cost[N0 + j] = 1.0;
    }
    __syncthreads();

    // ---- BoundedSimplex::run
    for (int k = tid; k < N1; k += NT1) {
        nonbasic[k] = (k >= N0) ? 0 : 1;
        range[k] = hi[k] - lo[k];
        double s = 0.0;
        for (int r = 0; r < M0; ++r) s += A1[(size_t)(r) * N1 + k] * A1[(size_t)(r) * N1 + k];
        colnorm[k] = sqrt(s);
        x[k] = S1[k] == SSQP_UP ? hi[k] : lo[k];
    }
    __syncthreads();
    // (the column of A1 / of Y a thread works on is pulled into registers first when M0 <= 16: the sums below are
    //  sequential by construction, and with a dependent global load per term each term would cost a cache round trip)
    constexpr int MC = 12;
    constexpr bool FEW = !BIG;   // (launch_phase1 picks the build by M0 <= 12 = MC = LUC: the other build's paths are not compiled in)
    // The host keeps Y = invB * A[:, nonbasic] (Simplex.jl:595) and reads it in two places: the pricing (Y[:,k] . c[basis])
    // and the xb sum (Y[:,k] x_k of the nonbasic columns at a nonzero bound).  Here Y[:,k] only ever exists in the
    // registers of the thread that owns column k: the pricing's dot product is formed at once (same order r = 0 .. M0 - 1
    // as the host's price()) and stored -- one number per column instead of M0 written and read back -- and the few
    // columns the xb sum needs are formed again where it needs them (same sums, same order: same bits).
    auto refreshY = [&]() {
        if (tid < M0) acc[tid] = basis[tid] >= N0 ? 1.0 : 0.0;  // c[basis] (LDS: broadcast reads below)
        __syncthreads();
        // slack and artificial columns are (signed) unit vectors: (invB * a)_r is the sum's single nonzero term -- the same
        // bits as the sum (its other terms are products with 0.0 and leave the value, or +0.0, as it is)
        for (int e = tid; e < J + M0; e += NT1) {
            const int k = e < J ? N + e : N0 + (e - J);
            if (!nonbasic[k]) continue;
            const int row = e < J ? M + e : e - J;
            const double val = e < J ? 1.0 : sgnArt[row];
            double sd = 0.0;
            if (FEW) {
                double iv[MC], cb[MC];
#pragma unroll
                for (int r = 0; r < MC; ++r) {
                    const int rr = r < M0 ? r : 0;
                    iv[r] = invB[row * M0 + rr];
                    cb[r] = acc[rr];
                }
#pragma unroll
                for (int r = 0; r < MC; ++r)
                    if (r < M0) {
                        const double sr = 0.0 + iv[r] * val;
                        sd += sr * cb[r];
                    }
            } else {
                for (int r = 0; r < M0; ++r) {
                    const double sr = 0.0 + invB[(size_t)row * M0 + r] * val;
                    sd += sr * acc[r];
                }
            }
            sdot[k] = sd;
        }
        auto general = [&](int k) { return k < N1 && !(k >= N && (k < N + J || k >= N0)); };  // (unit columns: done above)
        if (FEW) {
            // two columns per thread and round (k, k + NT1): the entries of invB a row's sums need are read ONCE for both --
            // LDS reads, not arithmetic, are what sixteen wavefronts per CU queue for here -- and TOGETHER, from clamped
            // addresses (a read under the "t < M0" guard waits for its own round trip, eleven in a row)
            for (int k = tid; k < N1; k += 2 * NT1) {
                const int kb = k + NT1;
                const bool doA = general(k) && nonbasic[k], doB = general(kb) && nonbasic[kb < N1 ? kb : k];
                if (!doA && !doB) continue;
                const double *akA = A1 + (doA ? k : kb), *akB = A1 + (doB ? kb : k);  // (entry t of a column: ak[t * N1])
                double avA[MC], avB[MC];
#pragma unroll
                for (int t = 0; t < MC; ++t) {
                    const size_t o = (size_t)(t < M0 ? t : 0) * N1;
                    avA[t] = akA[o];
                    avB[t] = akB[o];
                }
                double sdA = 0.0, sdB = 0.0;
                for (int r = 0; r < M0; ++r) {
                    double iv[MC];
#pragma unroll
                    for (int t = 0; t < MC; ++t) iv[t] = invB[(t < M0 ? t : 0) * M0 + r];
                    const double a0 = acc[r];
                    double sA = 0.0, sB = 0.0;
#pragma unroll
                    for (int t = 0; t < MC; ++t)
                        if (t < M0) {
                            sA += iv[t] * avA[t];
                            sB += iv[t] * avB[t];
                        }
                    sdA += sA * a0;
                    sdB += sB * a0;
                }
                if (doA) sdot[k] = sdA;
                if (doB) sdot[kb] = sdB;
            }
        } else {
            // Many rows (M0 > 12; cfg5: 72): only the rows whose basic variable is ARTIFICIAL count -- c[basis[r]] is 0 for the
            // others, and s * 0.0 = +-0.0 changes no bit of a sum that started at +0.0 and so is never -0.0 (only an Inf / NaN
            // row sum would have left a NaN behind) -- and they go SIX at a time over FOUR columns per thread: 24 independent
            // accumulation chains, every entry of the LP fetched once per six rows and every entry of inv(B) once per four
            // columns (the first version walked one column with four rows and a dependent global load per term: 6.2 M cycles
            // per refresh at M0 = 72, 80 % of cfg5's Phase-1).  Per element the sums keep the host's order: t ascending inside
            // a row, rows ascending.
#ifdef SSQP_PHASE_PROFILE
            unsigned long long ySet = 0, yLoop = 0, yEpi = 0;
#endif
            int *art = piv;   // (the LU's integer scratch is idle here)
            const int nArt = compact_columns(tid, M0, art, &misc[0], [&](int r) { return basis[r] >= N0; });
            constexpr int RG = BIG ? 8 : 6, CG = 4;
            // inv(B) of a simplex basis is SPARSE while unit columns (slacks, artificials) make up most of B -- cfg5: 3 % to
            // 30 % of its entries are nonzero over the first 150 of 273 basis changes -- and a term inv(B)[r, t] * a with an
            // exact zero factor adds +-0.0 to a sum that is never -0.0: leaving it out changes no bit (finite LP data).  Per
            // group of RG rows the steps t with a nonzero entry in ANY of them are listed once per refresh (one wavefront per
            // group, ballot order = ascending t); the column loop below walks the list instead of 0 .. M0 - 1.
            int *tl = reinterpret_cast<int *>(terms);   // (the xb sum's staging, idle here: group g at g (M0 + 1): count, then the steps)
            double *IV = Bm;                             // (the LU's scratch, idle here: the listed steps' entries of a group's RG rows,
                                                         //  RG contiguous doubles per step, group g at g M0 RG)
            {
                const int wv = tid >> 6, lane = tid & 63;
                for (int g = wv; g * RG < nArt; g += NT1 / 64) {
                    int *tg = tl + g * (M0 + 1);
                    double *ivg = IV + (size_t)g * M0 * RG;
                    int base = 0;
                    for (int tb = 0; tb < M0; tb += 64) {
                        const int t = tb + lane;
                        int rows = 0;   // bit v: row v of the group has a nonzero entry at step t
                        double e[RG];
#pragma unroll
                        for (int v = 0; v < RG; ++v) e[v] = 0.0;
                        if (t < M0) {
#pragma unroll
                            for (int v = 0; v < RG; ++v) e[v] = invB[(size_t)t * M0 + art[g * RG + v < nArt ? g * RG + v : g * RG]];
#pragma unroll
                            for (int v = 0; v < RG; ++v) rows |= (g * RG + v < nArt && e[v] != 0.0) ? 1 << v : 0;
                        }
                        const bool f = rows != 0;
                        const unsigned long long m = __ballot(f);
                        if (f) {
                            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                            tg[1 + pos] = t | (rows << 8);
#pragma unroll
                            for (int v = 0; v < RG; ++v) ivg[(size_t)pos * RG + v] = e[v];
                        }
                        base += __popcll(m);
                    }
                    if (lane == 0) tg[0] = base;
                }
            }
            __syncthreads();
            for (int kb = 0; kb < N1; kb += CG * NT1) {
                int kc[CG];
                bool on[CG];
                bool anyOn = false;
#pragma unroll
                for (int u = 0; u < CG; ++u) {
                    const int k = kb + tid + u * NT1;
                    on[u] = general(k) && nonbasic[k < N1 ? k : 0];
                    kc[u] = on[u] ? k : 0;   // (a column that is not wanted reads column 0: its sums are dropped)
                    anyOn = anyOn || on[u];
                }
                double sd[CG];
#pragma unroll
                for (int u = 0; u < CG; ++u) sd[u] = 0.0;
                if (anyOn) {
                    for (int g0 = 0; g0 < nArt; g0 += RG) {
#ifdef SSQP_PHASE_PROFILE
                        const unsigned long long yA = __builtin_amdgcn_s_memtime();
                        unsigned long long yB = yA, yC = yA;
#endif
                        const int *tg = tl + (g0 / RG) * (M0 + 1);
                        const int nT = tg[0];
                        int rr[RG];
                        double wt[RG];
#pragma unroll
                        for (int v = 0; v < RG; ++v) {
                            rr[v] = art[g0 + v < nArt ? g0 + v : g0];
                            wt[v] = g0 + v < nArt ? 1.0 : 0.0;   // (a group's unused places repeat its first row with weight 0)
                        }
                        double sacc[RG][CG];
#pragma unroll
                        for (int v = 0; v < RG; ++v)
#pragma unroll
                            for (int u = 0; u < CG; ++u) sacc[v][u] = 0.0;
                        if (nT > 0) {
                            // (the LP's entries of step i + 2 are requested before step i's products are formed, the step
                            //  numbers another step ahead: with one workgroup on the chip nobody else hides the L2 round trip)
                            // (list entries: step | rows << 8 -- a step adds to the sums of the rows that have an entry there and
                            //  to no others: the diagonal of an artificial row is a step of its own, nonzero in that row alone.
                            //  What bounds this loop is instruction issue -- 64 double-precision operations per full step, and
                            //  as many again for addresses, copies and waits in its first version -- so: the rows' entries come
                            //  packed (RG contiguous doubles per listed step, one address), the LP's entries through a scalar
                            //  row base plus the thread's column offset, in BLOCKS of four steps requested one block ahead, and
                            //  the two register sets change roles instead of being copied)
#ifndef P1_Y_SB
#define P1_Y_SB 4
#endif
                            constexpr int SB = P1_Y_SB;
                            const double *ivp = IV + (size_t)(g0 / RG) * M0 * RG;
                            int ea[SB], eb[SB];
                            double bufA[SB][CG], bufB[SB][CG], iv[RG];
#pragma unroll
                            for (int q = 0; q < SB; ++q) ea[q] = uni(tg[1 + (q < nT ? q : nT - 1)]);
#pragma unroll
                            for (int q = 0; q < SB; ++q) {
                                const double *rowp = A1 + (size_t)(ea[q] & 255) * N1;
#pragma unroll
                                for (int u = 0; u < CG; ++u) bufA[q][u] = rowp[kc[u]];
                            }
#pragma unroll
                            for (int v = 0; v < RG; ++v) iv[v] = ivp[v];
                            auto block = [&](double (&cu)[SB][CG], double (&nx)[SB][CG], int (&ecu)[SB], int (&enx)[SB], int ib) {
#pragma unroll
                                for (int q = 0; q < SB; ++q) enx[q] = uni(tg[1 + (ib + SB + q < nT ? ib + SB + q : nT - 1)]);
#pragma unroll
                                for (int q = 0; q < SB; ++q) {
                                    const double *rowp = A1 + (size_t)(enx[q] & 255) * N1;
#pragma unroll
                                    for (int u = 0; u < CG; ++u) nx[q][u] = rowp[kc[u]];
                                }
#pragma unroll
                                for (int q = 0; q < SB; ++q) {
                                    if (ib + q < nT) {
                                        double ivn[RG];
                                        const double *nextp = ivp + (size_t)(ib + q + 1 < nT ? ib + q + 1 : ib + q) * RG;
#pragma unroll
                                        for (int v = 0; v < RG; ++v) ivn[v] = nextp[v];
                                        int rows = ecu[q] >> 8;
#ifdef P1_Y_FULLKIND
                                        // (a step most rows take part in runs all of them without the eight tests: a product with
                                        //  an exact zero adds +-0.0 and changes no bit; only the steps of one or two rows -- the
                                        //  diagonals of the artificial rows -- pay for being picked out)
                                        if (__builtin_popcount(rows) >= P1_Y_FULLKIND) rows = (1 << RG) - 1;
                                        if (rows == (1 << RG) - 1) {
#pragma unroll
                                            for (int v = 0; v < RG; ++v) {
                                                double p0 = iv[v] * cu[q][0], p1 = iv[v] * cu[q][1], p2 = iv[v] * cu[q][2], p3 = iv[v] * cu[q][3];
                                                asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                                                sacc[v][0] += p0;
                                                sacc[v][1] += p1;
                                                sacc[v][2] += p2;
                                                sacc[v][3] += p3;
                                            }
                                        } else
#endif
#pragma unroll
                                        for (int v = 0; v < RG; ++v)
                                            if (rows & (1 << v)) {
                                                // (the four products first, then the four sums: back to back, every sum would
                                                //  wait out its own product's latency -- there is no second wavefront on the SIMD)
                                                static_assert(CG == 4, "the product / sum interleave below is written for four columns");
                                                double p0 = iv[v] * cu[q][0], p1 = iv[v] * cu[q][1], p2 = iv[v] * cu[q][2], p3 = iv[v] * cu[q][3];
                                                asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                                                sacc[v][0] += p0;
                                                sacc[v][1] += p1;
                                                sacc[v][2] += p2;
                                                sacc[v][3] += p3;
                                            }
#pragma unroll
                                        for (int v = 0; v < RG; ++v) iv[v] = ivn[v];
                                    }
                                }
                            };
#ifdef SSQP_PHASE_PROFILE
                            yB = __builtin_amdgcn_s_memtime();
#endif
                            for (int ib = 0; ib < nT; ib += 2 * SB) {
                                block(bufA, bufB, ea, eb, ib);
                                if (ib + SB < nT) block(bufB, bufA, eb, ea, ib + SB);
                            }
#ifdef SSQP_PHASE_PROFILE
                            yC = __builtin_amdgcn_s_memtime();
#endif
                        }
#pragma unroll
                        for (int v = 0; v < RG; ++v)
#pragma unroll
                            for (int u = 0; u < CG; ++u) sd[u] += sacc[v][u] * wt[v];
#ifdef SSQP_PHASE_PROFILE
                        {
                            asm volatile("" : "+v"(sd[0]));
                            const unsigned long long yD = __builtin_amdgcn_s_memtime();
                            ySet += yB - yA;
                            yLoop += yC - yB;
                            yEpi += yD - yC;
                        }
#endif
                    }
                }
#pragma unroll
                for (int u = 0; u < CG; ++u)
                    if (on[u]) sdot[kc[u]] = sd[u];
            }
#ifdef SSQP_PHASE_PROFILE
            if (tid == 0) {
                (void)__hip_atomic_fetch_add(&g_p1phase[9], ySet, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                (void)__hip_atomic_fetch_add(&g_p1phase[12], yLoop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                (void)__hip_atomic_fetch_add(&g_p1phase[13], yEpi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#endif
        }
        __syncthreads();
    };
    refreshY();
    P1_STAMP(0);  // set-up
    int status = 1;
    long loop = 0;
    for (;;) {
        // price: signed reduced costs; the entering candidate
        loop += 1;
        // The reference's loop has no limit (Simplex.jl:486); Bland's rule ends it after N1 passes on any finite LP,
        // but a workgroup that never leaves hangs the stream for good: an LP still pivoting after 64 N1 + 1024 passes
        // (NaN-poisoned or cycling in floating point) is given up as a numerical error (status -1)
        if (loop > 64l * N1 + 1024) {
            status = -1;
            break;
        }
        P1_COUNT(14);
        const bool bland = loop > N1;
        double best = -INF;
        int bidx = 0x7fffffff;
        for (int k = tid; k < N1; k += NT1) {
            // (everything the column needs is requested at once, basic or not: one memory round trip per column)
            const int nb = nonbasic[k];
            const int sk = S1[k];
            const double ck = k >= N0 ? 1.0 : 0.0;  // (= cost[k]: the Phase-1 objective is the sum of the artificials)
            const double nk = colnorm[k];
            const double s = sdot[k];
            if (!nb) continue;
            double hv = ck - s;
            if (sk == SSQP_DN) hv = -hv;
            if (hv > tol) {
                const double v = bland ? 0.0 : hv / nk;
                if (v > best || (v == best && k < bidx)) best = v, bidx = k;  // (per thread: ascending k, first max)
            }
        }
        P1_STAMP(1);  // pricing
        block_first_max<NT1>(best, bidx, redv, redi);
        P1_STAMP(2);  // block maximum
        if (bidx == 0x7fffffff) break;  // no improving candidate: optimal
        const int k = bidx;
        const double *ak = A1 + k;  // (entry t of column k: ak[t * N1])
        if (!FEW) {  // (many rows: the column goes to LDS in one round trip -- the LU's scratch is idle -- instead of eight entries per trip)
            for (int t = tid; t < M0; t += NT1) Bm[t] = ak[(size_t)t * N1];
            __syncthreads();
        }
        for (int r = tid; r < M0; r += NT1) {
            double s = 0.0;
            if (FEW) {  // (the column's entries and the row of invB are requested together: one round trip)
                double av[MC], iv[MC];
#pragma unroll
                for (int t = 0; t < MC; ++t) {
                    const int tt = t < M0 ? t : 0;
                    av[t] = ak[(size_t)tt * N1];
                    iv[t] = invB[tt * M0 + r];
                }
#pragma unroll
                for (int t = 0; t < MC; ++t)
                    if (t < M0) s += iv[t] * av[t];
            } else {  // (eight terms per memory round trip, added in order)
                for (int t0 = 0; t0 < M0; t0 += 8) {
                    double iv8[8], av8[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int t = t0 + q < M0 ? t0 + q : t0;
                        iv8[q] = invB[(size_t)t * M0 + r];
                        av8[q] = Bm[t];
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (t0 + q < M0) s += iv8[q] * av8[q];
                }
            }
            pv[r] = s;
        }
        {   // the candidate ratio of every basic row by the thread that formed its pv (the host's expression, so the same
            // bits); thread 0 then only SELECTS in row order -- no division and no memory round trip in its chain
            const bool fromLower = S1[k] == SSQP_DN;
            if (tid < M0) {
                const int i = basis[tid];
                const double bl = lo[i], bh = hi[i];
                const double p = pv[tid];  // (written by this thread above)
                const bool pos = p > tol, neg = p < -tol;
                const bool toLower = fromLower ? pos : neg;
                blo[tid] = (pos || neg) ? (xb[tid] - (toLower ? bl : bh)) / p : 0.0;
                piv[tid] = (pos || neg) ? (toLower ? 1 : 2) : 0;
                acc[tid] = bl;  // (the bounds of the basic variables by row: the leaving one's new value is one of them)
                bhi[tid] = bh;
            }
        }
        // (what thread 0 needs of the entering column after the barrier, requested before it)
        double hiK = 0.0, loK = 0.0, rangeK = 0.0;
        if (tid == 0) {
            hiK = hi[k];
            loK = lo[k];
            rangeK = range[k];
        }
        __syncthreads();
        P1_STAMP(3);  // entering column
        // ratio test (first minimum / first maximum over the basic rows, in row order).  Many rows: the first wavefront reduces
        // (value, row) pairs -- candidates first, ties to the smaller row, as in the wavefront kernel -- instead of one thread
        // walking the rows (21 k cycles per pass at 72 rows)
        int rtM = 0, rtRow = 0, rtTo = SSQP_DN;
        double rtLr = 0.0;
        if (!FEW && tid < 64) {
            const bool fromLower = S1[k] == SSQP_DN;
            KeyMin cur{INF, 0x7fffffff};
            bool anyc = false;
            for (int j = tid; j < M0; j += 64) {
                const int code = piv[j];
                const double ratio = blo[j];
                const bool cand = code != 0;
                cur = keymin(cur, KeyMin{cand ? (fromLower ? ratio : -ratio) : INF, cand ? j : j + 4096});
                anyc = anyc || cand;
            }
            const bool any = __ballot(anyc) != 0ull;
            const KeyMin kr = wave_keymin(cur);
            rtM = any ? 1 : 0;
            rtRow = any ? kr.ord : 0;
            rtLr = fromLower ? kr.v : -kr.v;
            rtTo = piv[rtRow] == 1 ? SSQP_DN : SSQP_UP;
        }
        if (tid == 0) {
            const bool fromLower = S1[k] == SSQP_DN;
            int m = rtM, li = -1;
            double lr = rtLr;
            int lrow = rtRow, lto = rtTo;
            auto consider = [&](int j, int code, double ratio) {
                if (code == 0) return;
                const bool better = (m == 0) || (fromLower ? (ratio < lr) : (ratio > lr));
                if (better) lr = ratio, li = m, lrow = j, lto = (code == 1) ? SSQP_DN : SSQP_UP;
                ++m;
            };
            if (FEW) {
                double rt[MC];
                int cd[MC];
#pragma unroll
                for (int j = 0; j < MC; ++j) {
                    const int jj = j < M0 ? j : 0;
                    rt[j] = blo[jj];
                    cd[j] = piv[jj];
                }
#pragma unroll
                for (int j = 0; j < MC; ++j)
                    if (j < M0) consider(j, cd[j], rt[j]);
            }
            int action = 0, leaveStatus = SSQP_DN, st = 0;  // st: 3 = unbounded
            if (fromLower) {
                const bool finiteUp = hiK < INF;
                if (m == 0) {
                    if (!finiteUp) st = 3;
                    else action = -1;
                } else {
                    if (finiteUp && lr >= rangeK) action = -1;
                    else {
                        if (!finiteUp && isinf(lr)) st = 3;
                        else action = lrow + 1, leaveStatus = lto;
                    }
                }
            } else {
                if (m == 0) action = -2;
                else if (lr <= -rangeK) action = -2;
                else action = lrow + 1, leaveStatus = lto;
            }
            (void)li;
            redv[BIG ? 8 : 3] = (leaveStatus == SSQP_DN) ? acc[lrow] : bhi[lrow];  // (a pivot: the bound the leaving variable goes to)
            misc[2] = action;
            misc[3] = leaveStatus;
            misc[1] = st;
        }
        __syncthreads();
        P1_STAMP(4);  // ratio test
        if (misc[1] == 3) {
            status = 3;
            break;
        }
        const int action = misc[2];
        if (action == -1) {
            if (tid == 0) S1[k] = SSQP_UP, x[k] = hiK;
        } else if (action == -2) {
            if (tid == 0) S1[k] = SSQP_DN, x[k] = loK;
        } else {
            if (FEW) {
                if (tid == 0) {
                    const int leaving = basis[action - 1];
                    misc[6] = leaving;
                    nonbasic[k] = 0;
                    nonbasic[leaving] = 1;
                    basis[action - 1] = k;
                    // sort(basis) in registers (the basis was sorted before the exchange: one pass each way)
                    int bs[MC];
#pragma unroll
                    for (int a2 = 0; a2 < MC; ++a2) bs[a2] = (a2 < M0) ? basis[a2 < M0 ? a2 : 0] : 0x7fffffff;
#pragma unroll
                    for (int a2 = 0; a2 + 1 < MC; ++a2) {  // the new entry sinks up ...
                        const int lo2 = min(bs[a2], bs[a2 + 1]), hi2 = max(bs[a2], bs[a2 + 1]);
                        bs[a2] = lo2;
                        bs[a2 + 1] = hi2;
                    }
#pragma unroll
                    for (int a2 = MC - 2; a2 >= 0; --a2) {  // ... or down
                        const int lo2 = min(bs[a2], bs[a2 + 1]), hi2 = max(bs[a2], bs[a2 + 1]);
                        bs[a2] = lo2;
                        bs[a2 + 1] = hi2;
                    }
#pragma unroll
                    for (int a2 = 0; a2 < MC; ++a2)
                        if (a2 < M0) basis[a2] = bs[a2];
                }
            } else if (tid < 64) {
                // sort(basis) after the exchange: the basis was sorted, one entry leaves (row m), k enters at its rank p among the
                // others -- every lane computes where its entries of the new order come from (M0 <= 128: two per lane); one
                // thread's insertion sort was 20 k cycles of dependent LDS round trips at 72 rows
                const int mrow = action - 1;
                int o[2];
                unsigned long long below[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = tid + 64 * h;
                    o[h] = i < M0 ? basis[i] : 0x7fffffff;
                    below[h] = __ballot(i < M0 && i != mrow && o[h] < k);
                }
                const int pIns = __popcll(below[0]) + __popcll(below[1]);
                const int leaving = basis[mrow];
                int nw[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = tid + 64 * h;
                    const int j = i < pIns ? i : i - 1;                 // index among the entries that stay
                    const int src = j < mrow ? j : j + 1;               // ... and in the old order
                    nw[h] = (i == pIns) ? k : basis[i < M0 && i != pIns ? src : 0];
                }
                wave_order();   // (every read of the old order is through)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if (tid + 64 * h < M0) basis[tid + 64 * h] = nw[h];
                if (tid == 0) {
                    misc[6] = leaving;
                    nonbasic[k] = 0;
                    nonbasic[leaving] = 1;
                }
            }
            __syncthreads();
            // (eight elements' column numbers and entries are requested together)
            for (int e0 = tid; e0 < M0 * M0; e0 += 8 * NT1) {
                double gv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = e0 + q * NT1 < M0 * M0 ? e0 + q * NT1 : e0;
                    gv[q] = A1[(size_t)(e % M0) * N1 + basis[e / M0]];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (e0 + q * NT1 < M0 * M0) invB[e0 + q * NT1] = gv[q];
            }
            __syncthreads();
            P1_STAMP(5);  // basis sort + gather
            P1_COUNT(15);
            bool okLu;
            if (FEW) {  // (one wavefront, the matrix in registers; the others wait for its verdict)
                if (tid < 64) {
                    const bool okw = invert_lu_cols(invB, Bm, piv, M0);
                    if (tid == 0) misc[1] = okw ? 1 : 0;
                }
                __syncthreads();
                okLu = misc[1] != 0;
            } else {
                okLu = invert_lu<NT1>(tid, invB, Bm, terms, piv, M0, &misc[1]);   // (terms: idle here, >= 2 M0 doubles)
            }
            if (!okLu) {  // lu() of the reference throws (Simplex.jl:590)
                status = -1;
                break;
            }
            if (tid == 0) {
                const int leaving = misc[6], leaveStatus = misc[3];
                S1[k] = SSQP_IN;
                S1[leaving] = leaveStatus;
                x[leaving] = redv[BIG ? 8 : 3];
            }
            P1_STAMP(6);  // inv(lu(B))
            refreshY();
            P1_STAMP(7);  // Y = invB A
        }
        __syncthreads();
        // xb = invB*b - Y*x[nonbasic]: the nonbasic columns at a nonzero value, ascending
        {
            auto atBound = [&](int kk) { return nonbasic[kk] && x[kk] != 0.0; };
            const int cnt = FEW ? compact_columns(tid, N1, list, &misc[0], atBound)
                                : compact_columns_wg<NT1>(tid, N1, list, reinterpret_cast<int *>(terms), atBound);
            // the sum runs over the listed columns in ascending order, one rounded multiply and one rounded add per term
            // (the host's order): the PRODUCTS of a chunk of columns are formed by all threads at once (one memory round
            // trip for the chunk instead of one per term), the adds stay in order
            double a2 = 0.0;
            if (!FEW) {
                // Many rows: the listed columns of the LP are STAGED in LDS a chunk at a time (the LU's scratch, idle here) --
                // a (column, row) element per thread walking its 72 terms through L2 was nine dependent round trips, 200 k
                // cycles per pass on cfg5 -- and thread (g, r) forms row r of the columns g, g + G, ... of the chunk: one entry
                // of inv(B) is read per step for up to XBC columns, the LP's entries are broadcast reads.  Per element the
                // sum runs over t ascending, as refreshY forms Y[r, k].
                constexpr int XBC = XB_CHUNK_BIG / 4;                              // (XB_CHUNK_BIG = 40 columns = 4 groups x 10)
                const int G = 4;                                                   // column groups: 4 M0 <= 512 threads (M0 <= 128)
                const int chunk = XB_CHUNK_BIG < M0 ? XB_CHUNK_BIG : M0;           // (chunk x M0 doubles fit the scratch)
                const int g = tid / M0, r = tid - g * M0;
                double *Ac = Bm;                                                   // column t of the chunk at t * M0
                for (int t0 = 0; t0 < cnt; t0 += chunk) {
                    const int nt = cnt - t0 < chunk ? cnt - t0 : chunk;
                    for (int e = tid; e < nt * M0; e += NT1) {
                        const int t = e / M0, t2 = e - t * M0;
                        Ac[e] = A1[(size_t)t2 * N1 + list[t0 + t]];
                    }
                    __syncthreads();
                    if (g < G) {
                        double y[XBC];
#pragma unroll
                        for (int c = 0; c < XBC; ++c) y[c] = 0.0;
                        const int nc = (nt - g + G - 1) / G;                       // this thread's columns: g + c G < nt
                        for (int t2 = 0; t2 < M0; t2 += 2) {                       // (two steps' reads go out together)
                            const int t2b = t2 + 1 < M0 ? t2 + 1 : t2;
                            const double iv0 = invB[(size_t)t2 * M0 + r], iv1 = invB[(size_t)t2b * M0 + r];
                            double a0[XBC], a1[XBC];
#pragma unroll
                            for (int c = 0; c < XBC; ++c) {
                                const int t = c < nc ? g + c * G : g < nt ? g : 0;
                                a0[c] = Ac[(size_t)t * M0 + t2];
                                a1[c] = Ac[(size_t)t * M0 + t2b];
                            }
#pragma unroll
                            for (int c = 0; c < XBC; ++c) y[c] += iv0 * a0[c];
                            if (t2 + 1 < M0) {
#pragma unroll
                                for (int c = 0; c < XBC; ++c) y[c] += iv1 * a1[c];
                            }
                        }
#pragma unroll
                        for (int c = 0; c < XBC; ++c)
                            if (c < nc) terms[(size_t)(g + c * G) * M0 + r] = y[c] * x[list[t0 + g + c * G]];
                    }
                    __syncthreads();
                    if (tid < M0) {  // (eight terms per LDS round trip, added in order)
                        for (int t0b = 0; t0b < nt; t0b += 8) {
                            double tv[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) tv[q] = terms[(size_t)(t0b + q < nt ? t0b + q : t0b) * M0 + tid];
#pragma unroll
                            for (int q = 0; q < 8; ++q)
                                if (t0b + q < nt) a2 += tv[q];
                        }
                    }
                    __syncthreads();
                }
            } else
            for (int t0 = 0; t0 < cnt; t0 += XB_CHUNK) {
                const int nt = cnt - t0 < XB_CHUNK ? cnt - t0 : XB_CHUNK;
                // (M0 <= 16: thread -> (column t = tid / 16 + 16 i, row r = tid % 16), no integer division per element)
                const int eStep = M0 <= 16 ? 16 : NT1, eEnd = M0 <= 16 ? nt : nt * M0;
                for (int e = M0 <= 16 ? (tid >> 4) : tid; e < eEnd; e += eStep) {
                    int t, r;
                    if (M0 <= 16) {
                        t = e;
                        r = tid & 15;
                        if (r >= M0) break;
                    } else {
                        t = e / M0;
                        r = e - t * M0;
                    }
                    const int kk = list[t0 + t];
                    double y = 0.0;  // Y[r, kk] = (invB * A1[:, kk])_r, as refreshY forms it
                    if (FEW) {
                        double av[MC], iv[MC];
#pragma unroll
                        for (int t2 = 0; t2 < MC; ++t2) {
                            const int tt = t2 < M0 ? t2 : 0;
                            av[t2] = A1[(size_t)tt * N1 + kk];
                            iv[t2] = invB[tt * M0 + r];
                        }
#pragma unroll
                        for (int t2 = 0; t2 < MC; ++t2)
                            if (t2 < M0) y += iv[t2] * av[t2];
                    } else {  // (eight terms per memory round trip, added in order)
                        for (int t0 = 0; t0 < M0; t0 += 8) {
                            double iv8[8], av8[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const int t2 = t0 + q < M0 ? t0 + q : t0;
                                iv8[q] = invB[(size_t)t2 * M0 + r];
                                av8[q] = A1[(size_t)t2 * N1 + kk];
                            }
#pragma unroll
                            for (int q = 0; q < 8; ++q)
                                if (t0 + q < M0) y += iv8[q] * av8[q];
                        }
                    }
                    terms[(size_t)t * M0 + r] = y * x[kk];
                }
                __syncthreads();
                if (tid < M0) {  // (eight terms per LDS round trip, added in order)
                    for (int t0b = 0; t0b < nt; t0b += 8) {
                        double tv[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) tv[q] = terms[(size_t)(t0b + q < nt ? t0b + q : t0b) * M0 + tid];
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (t0b + q < nt) a2 += tv[q];
                    }
                }
                __syncthreads();
            }
            for (int r = tid; r < M0; r += NT1) {
                double s = 0.0;
                if (FEW) {
                    double iv[MC], rv[MC];
#pragma unroll
                    for (int t = 0; t < MC; ++t) {
                        const int tt = t < M0 ? t : 0;
                        iv[t] = invB[tt * M0 + r];
                        rv[t] = rhs[tt];
                    }
#pragma unroll
                    for (int t = 0; t < MC; ++t)
                        if (t < M0) s += iv[t] * rv[t];
                } else {
                    for (int t0 = 0; t0 < M0; t0 += 8) {
                        double iv8[8], rv8[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int t = t0 + q < M0 ? t0 + q : t0;
                            iv8[q] = invB[(size_t)t * M0 + r];
                            rv8[q] = rhs[t];
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (t0 + q < M0) s += iv8[q] * rv8[q];
                    }
                }
                xb[r] = s - a2;
            }
            __syncthreads();
        }
        P1_STAMP(8);  // xb
    }
    __syncthreads();
    P1_FLUSH();
    // ---- finish(): values of the basic variables; then initQP's mapping back (SSQP.jl:533-557)
    if (status >= 0)
        for (int j = tid; j < M0; j += NT1) x[basis[j]] = xb[j];
    __syncthreads();
    for (int k = tid; k < N; k += NT1) x0[k] = x[k];
    for (int k = tid; k < N + J; k += NT1) S[k] = S1[k];
    __syncthreads();
    auto endHere = [&](int st) {  // (single-launch solveQP(Q): the QP ends with Phase-1's verdict)
        if (!P.hoList) return;
        double *z = P.z + (size_t)prob * N;
        for (int k = tid; k < N; k += NT1) z[k] = x[k];
        if (tid == 0) {
            P.status64[prob] = st;
            if (P.detail) P.detail[prob] = st < 0 ? SSQP_DETAIL_SINGULAR_LU : SSQP_DETAIL_NONE;
        }
    };
    if (P.hoList && P.stats && tid == 0) {
        ssqp_stats z0;
        z0.iters = 0; z0.alg_bytes = 0; z0.read_bytes = 0; z0.alg_flops = 0; z0.sum_k3 = 0; z0.max_k = 0; z0.path = 128;
        P.stats[prob] = z0;
    }
    if (status < 0) {
        if (tid == 0) P.status[prob] = -1;
        endHere(-1);
        return;
    }
    if (tid == 0) {
        double art = 0.0;
        for (int k = N0; k < N1; ++k) art += x[k];
        misc[0] = (art > tol) ? 0 : 1;
        P.status[prob] = misc[0];
    }
    __syncthreads();
    if (misc[0] == 0) {
        endHere(0);
        return;
    }
    for (int k = N + tid; k < N + J; k += NT1) S[k] = (S1[k] == SSQP_IN) ? SSQP_OE : SSQP_EO;
    for (int t = tid; t < nfree; t += NT1) {
        x0[freeVars[t]] = x[freeVars[t]] - x[N + J + t];
        S[freeVars[t]] = SSQP_IN;
    }
    __syncthreads();
    for (int t = tid; t < nup; t += NT1) x0[upperOnly[t]] = -x0[upperOnly[t]];  // (statuses stay: SSQP.jl:552-557 is a no-op)
    if (P.hoList) {  // on into the loop: a hand-over at pass 0
        __syncthreads();
        double *z = P.z + (size_t)prob * N;
        for (int k = tid; k < N; k += NT1) z[k] = x0[k];
        if (tid == 0) {
            P.hoIter[prob] = 0;
            const unsigned slot = atomicAdd(P.hoCount, 1u);
            P.hoList[slot] = prob;
        }
    }
}

template <bool BIG>
__global__ __launch_bounds__(NTB<BIG>, BIG ? 1 : 4) void ssqp_phase1_kernel(P1Params P) {  // (few rows: four workgroups per CU, 1,024 QPs resident)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // one QP per workgroup; on a list (what the wavefront kernel left) a bounded grid strides over it
    const int n = P.list ? (int)*P.listCount : P.nprob;
    for (int it = blockIdx.x; it < n; it += gridDim.x) {
        phase1_one_wg<BIG>(P, P.list ? P.list[it] : it, smem);
        __syncthreads();
    }
}

}  // namespace p1

size_t phase1_ws_doubles(int N, int M, int J) {
    const int M0 = M + J, N1 = 2 * N + J + M0;  // (every variable free: n = N)
    return (size_t)M0 * N1 + 7 * (size_t)N1 + 8;
}
size_t phase1_ws_ints(int N, int M, int J) {
    const int M0 = M + J, N1 = 2 * N + J + M0;
    return (size_t)2 * N + 3 * (size_t)N1 + 8;
}
size_t phase1_lds_bytes(int M, int J) {  // without the N1-vectors
    const size_t M0 = (size_t)(M + J);
    return (2 * M0 * M0 + (M0 > 12 ? 15 : 8) * M0 + (M0 > 12 ? 12 : 4) + (M0 > 12 ? p1::XB_CHUNK_BIG : p1::XB_CHUNK) * M0) * 8 +
           (3 * M0 + (M0 > 12 ? 8 : 4) + 8) * 4 + 64;
}
// with x, colnorm, sdot, S1, nonbasic of up to N1x = 2N + J + M0 columns in LDS
static size_t phase1_lds_bytes_vec(int N, int M, int J) {
    const size_t N1x = (size_t)2 * N + J + M + J;
    return phase1_lds_bytes(M, J) + N1x * (3 * 8 + 2 * 4) + 16;
}
hipError_t launch_phase1(int nprob, int N, int M, int J, const double *A, const double *G, const double *b, const double *g,
                         const double *d, const double *u, double tol, double *x0, int32_t *S, int32_t *status, double *ws,
                         size_t wsStride, int *wsInt, size_t wsIntStride, const unsigned int *listCount, const int *list, int gridCap,
                         const Phase1Handover *ho, hipStream_t stream) {
    p1::P1Params P;
    P.listCount = listCount; P.list = list;
    P.hoCount = ho ? ho->count : nullptr; P.hoList = ho ? ho->list : nullptr; P.hoIter = ho ? ho->iter : nullptr;
    P.z = ho ? ho->z : nullptr; P.status64 = ho ? ho->status : nullptr; P.detail = ho ? ho->detail : nullptr;
    P.stats = ho ? ho->stats : nullptr;
    P.nprob = nprob; P.N = N; P.M = M; P.J = J;
    P.A = A; P.G = G; P.b = b; P.g = g; P.d = d; P.u = u;
    P.tol = tol;
    P.x0 = x0; P.S = S; P.status = status;
    P.ws = ws; P.wsStride = wsStride; P.wsInt = wsInt; P.wsIntStride = wsIntStride;
    // four workgroups per CU keep 1,024 QPs resident: the N1-vectors go to LDS when they fit in a quarter of it
    size_t lds = phase1_lds_bytes(M, J);
    P.ldsVec = 0;
    // (on a list -- what the wavefront kernel left, usually nothing -- the launch stays small in every respect: a few
    //  blocks with the small LDS image find room on a busy chip at once instead of queueing behind another lane's kernels)
    if (!list && phase1_lds_bytes_vec(N, M, J) <= (size_t)LDS_BYTES / 4) {
        P.ldsVec = 1;
        lds = phase1_lds_bytes_vec(N, M, J);
    }
    static unsigned long long ldsSet = 0ull, ldsSetBig = 0ull;
    const int grid = (list && gridCap > 0 && gridCap < nprob) ? gridCap : nprob;
    if (M + J > 12) {   // many rows: the build with the wide refresh tiles
        hipError_t e = allow_full_lds(reinterpret_cast<const void *>(&p1::ssqp_phase1_kernel<true>), &ldsSetBig);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(p1::ssqp_phase1_kernel<true>, dim3(grid), dim3(p1::NTB<true>), lds, stream, P);
        return hipGetLastError();
    }
    hipError_t e = allow_full_lds(reinterpret_cast<const void *>(&p1::ssqp_phase1_kernel<false>), &ldsSet);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(p1::ssqp_phase1_kernel<false>, dim3(grid), dim3(p1::NT1), lds, stream, P);
    return hipGetLastError();
}

#ifdef SSQP_PHASE_PROFILE
int phase1_debug_phases(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(p1::g_p1phase), 24 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        static unsigned long long zero[24];
        if (hipMemcpyToSymbol(HIP_SYMBOL(p1::g_p1phase), zero, sizeof(zero)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

}  // namespace ssqp

#ifdef SSQP_PHASE_PROFILE
extern "C" int ssqp_debug_phase1_phases(unsigned long long *out16, int reset) { return ssqp::phase1_debug_phases(out16, reset); }
#endif
