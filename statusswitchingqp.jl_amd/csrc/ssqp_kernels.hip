// ssqp_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the active-set inner loop.
//
// One 256-thread workgroup (4 wavefronts) owns one QP and runs the WHOLE solveQP(Q,S,x0) loop
// (reference: src/SSQP.jl:237-377) in-kernel: no host round trip per iteration.  Workgroups are persistent
// (two per CU) and pull problem ids from a device counter.
//
// Per loop pass (names follow the reference; DESIGN.md section 4 has the long version):
//   compaction      F = (S .== IN), index lists                      SSQP.jl:276-289
//   factor sync     the LDL' factor of V[F,F] is KEPT across passes: append a released variable,
//                   delete a blocked one (replaces inv(cholesky(V[F,F])) from scratch, SSQP.jl:322)
//   E rows          bE = [b; g[Eg]] - AB*zB (kept for every row, follows zB by one column of [A;G] per status
//                   switch), X = [AE bE] by a K x W0 gather         SSQP.jl:290-295
//   rank filter     getRowsGJr(X, tol), operation for operation      utils.jl:49-86
//   c               hq = V[:,nz(zB)] zB + q kept in LDS, follows zB by one column of V per status switch;
//                   c = hq[F]                                        SSQP.jl:323-324
//   KKT solve       forward substitution of the border [AE' c] (only the appended row when nothing else
//                   changed), Schur system (AE V_FF^-1 AE') lambda = bE + AE V_FF^-1 c,
//                   one back substitution                            SSQP.jl:325-332
//   aStep!          ratio test as workgroup min-reduction            SSQP.jl:61-134
//   gamma pass      V[:,F] alpha + AB' alphaL (AXPY) + hq            SSQP.jl:351-352
//   KKTchk!         (value, order) argmin                            SSQP.jl:136-188
//   polishSz!                                                        SSQP.jl:10-32
//   freeK!          K == 0 pass                                      SSQP.jl:35-59
// plus the from-scratch path (pass 1 over V[:,F] with the V[F,F] gather fused in, panel LDL' of the
// bordered matrix) for the cases the kept-factor engine does not cover.
//
// The reference forms inv(cholesky(V[F,F])) and inv(cholesky(C)) explicitly; this kernel factors and
// solves (same KKT system).  Status decisions are threshold tests far above
// rounding noise, so S matches bit for bit away from exact ties; z/lambda agree to ~1e-13 relative.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssqp_hip.h"
#include "ssqp_internal.h"
#include "ssqp_device.h"

namespace ssqp {


// ---- diagnostic build only (-DSSQP_PHASE_PROFILE): cycles per phase, thread 0 of each workgroup ----
#ifdef SSQP_PHASE_PROFILE
__device__ unsigned long long g_phase[1024 * 32];
#define PHASE(C, n)                                                        \
    do {                                                                   \
        if (threadIdx.x == 0) {                                            \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();    \
            (void)__hip_atomic_fetch_add(&g_phase[(blockIdx.x & 1023) * 32 + (C).ph_cur], t_ - (C).ph_last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            (C).ph_last = t_;                                              \
            (C).ph_cur = (n);                                              \
        }                                                                  \
    } while (0)
#define SUBPHASE(slot, t0)                                                     \
    do {                                                                       \
        if (threadIdx.x == 0) {                                                \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();        \
            (void)__hip_atomic_fetch_add(&g_phase[(blockIdx.x & 1023) * 32 + (slot)], t_ - (t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            (t0) = t_;                                                         \
        }                                                                      \
    } while (0)
#define PCOUNT(slot)                                                           \
    do {                                                                       \
        if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(&g_phase[(blockIdx.x & 1023) * 32 + (slot)], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    } while (0)
#define SUBPHASE_DECL(t0) unsigned long long t0 = __builtin_amdgcn_s_memtime()
#define WSTAMP(slot, t0)                                                       \
    do {                                                                       \
        if ((threadIdx.x & 63) == 0) {                                         \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();        \
            (void)__hip_atomic_fetch_add(&g_phase[(blockIdx.x & 1023) * 32 + (slot)], t_ - (t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            (t0) = t_;                                                         \
        }                                                                      \
    } while (0)
#else
#define WSTAMP(slot, t0) do { } while (0)
#define PHASE(C, n) do { } while (0)
#define SUBPHASE(slot, t0) do { } while (0)
#define PCOUNT(slot) do { } while (0)
#define SUBPHASE_DECL(t0) do { } while (0)
#endif


struct Lds {
    double *z, *zm, *gam, *arena;
    double *hq;                          // cached hB + q (see the front half of iterate_kkt)
    double *bE, *aL, *tv, *dcol, *lin;  // MJ+1 each
    double *bEall;                       // b - [A;G] zB for EVERY row, valid together with hq (front half)
    double *red;                         // 2*NW
    int32_t *S;
    int *ired;                           // 2*NW + 8
    int16_t *pos, *idx, *perm, *rowsE, *ra, *iO, *fpos, *ordl;
    int16_t *ytag;                       // row id behind each kept border column (16 entries)
    double *evtz;                        // up to 16 (value, index) events of a pass: variables that became bound at a
    int *evti;                           //   nonzero value (hq and bEall are updated by their columns, see bound_event)
};

__device__ __forceinline__ double block_max(double v, const Lds &L) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_max(v);
    if (lane == 0) L.red[wave] = v;
    __syncthreads();
    double r = L.red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r = fmax(r, L.red[w]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_min(double v, const Lds &L) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_min(v);
    if (lane == 0) L.red[wave] = v;
    __syncthreads();
    double r = L.red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r = fmin(r, L.red[w]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ int block_or(int f, const Lds &L) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(f);
    if (lane == 0) L.ired[wave] = (m != 0ull);
    __syncthreads();
    int r = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) r |= L.ired[w];
    __syncthreads();
    return r;
}
__device__ __forceinline__ KeyMin block_keymin(KeyMin a, const Lds &L) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    a = wave_keymin(a);
    if (lane == 0) {
        L.red[wave] = a.v;
        L.ired[wave] = a.ord;
    }
    __syncthreads();
    KeyMin r{L.red[0], L.ired[0]};
#pragma unroll
    for (int w = 1; w < NW; ++w) r = keymin(r, KeyMin{L.red[w], L.ired[w]});
    __syncthreads();
    return r;
}

// packed bordered lower-triangular storage: column j holds rows j..R-1
__device__ __forceinline__ int coloff(int j, int R) { return j * R - ((j * (j - 1)) >> 1); }

// ------------------------------------------------------------ column streams
// Partial (per-lane) dot product of two V columns with the LDS vector w.
// VEC == 2: 16-byte loads, a full 1 KiB per wave instruction (needs N even).
template <int VEC>
__device__ __forceinline__ void dot2_cols(const double *__restrict__ c0, const double *__restrict__ c1,
                                          const double *w, int N, int lane, double &a0, double &a1) {
    a0 = 0.0;
    a1 = 0.0;
    if (VEC >= 2) {
#pragma unroll 4
        for (int r = lane * 2; r < N; r += 128) {
            const double2 v0 = *reinterpret_cast<const double2 *>(c0 + r);
            const double2 v1 = *reinterpret_cast<const double2 *>(c1 + r);
            const double2 ww = *reinterpret_cast<const double2 *>(w + r);
            a0 = fma(v0.x, ww.x, a0);
            a0 = fma(v0.y, ww.y, a0);
            a1 = fma(v1.x, ww.x, a1);
            a1 = fma(v1.y, ww.y, a1);
        }
    } else {
#pragma unroll 4
        for (int r = lane; r < N; r += 64) {
            const double ww = w[r];
            a0 = fma(c0[r], ww, a0);
            a1 = fma(c1[r], ww, a1);
        }
    }
}

// Same, and the rows that are free (pos >= kcol) are written into the packed
// factor: this is the V[F,F] gather of SSQP.jl:322 fused with c = V[B,F]'zB.
template <int VEC>
__device__ __forceinline__ void dot2_cols_gather(const double *__restrict__ c0, const double *__restrict__ c1,
                                                 const double *w, const int16_t *pos, double *fac, int off0,
                                                 int off1, int k0, int k1, bool two, int N, int lane,
                                                 double &a0, double &a1) {
    a0 = 0.0;
    a1 = 0.0;
    if (VEC >= 2) {
#pragma unroll 4
        for (int r = lane * 2; r < N; r += 128) {
            const double2 v0 = *reinterpret_cast<const double2 *>(c0 + r);
            const double2 v1 = *reinterpret_cast<const double2 *>(c1 + r);
            const double2 ww = *reinterpret_cast<const double2 *>(w + r);
            const int p0 = pos[r], p1 = pos[r + 1];
            a0 = fma(v0.x, ww.x, a0);
            a0 = fma(v0.y, ww.y, a0);
            a1 = fma(v1.x, ww.x, a1);
            a1 = fma(v1.y, ww.y, a1);
            if (p0 >= k0) fac[off0 + p0 - k0] = v0.x;
            if (p1 >= k0) fac[off0 + p1 - k0] = v0.y;
            if (two) {
                if (p0 >= k1) fac[off1 + p0 - k1] = v1.x;
                if (p1 >= k1) fac[off1 + p1 - k1] = v1.y;
            }
        }
    } else {
#pragma unroll 4
        for (int r = lane; r < N; r += 64) {
            const double ww = w[r];
            const double v0 = c0[r], v1 = c1[r];
            const int p0 = pos[r];
            a0 = fma(v0, ww, a0);
            a1 = fma(v1, ww, a1);
            if (p0 >= k0) fac[off0 + p0 - k0] = v0;
            if (two && p0 >= k1) fac[off1 + p0 - k1] = v1;
        }
    }
}

// Per-lane partial dot products of one contiguous global row (a constraint row of Ct) with two LDS vectors.
template <int VEC>
__device__ __forceinline__ void row_dot2(const double *__restrict__ row, const double *w1, const double *w2, int N,
                                         int lane, double &a1, double &a2) {
    a1 = 0.0;
    a2 = 0.0;
    if (VEC >= 2) {
#pragma unroll 4
        for (int r = lane * 2; r < N; r += 128) {
            const double2 v = *reinterpret_cast<const double2 *>(row + r);
            const double2 x = *reinterpret_cast<const double2 *>(w1 + r);
            const double2 y = *reinterpret_cast<const double2 *>(w2 + r);
            a1 = fma(v.y, x.y, fma(v.x, x.x, a1));
            a2 = fma(v.y, y.y, fma(v.x, y.x, a2));
        }
    } else {
#pragma unroll 4
        for (int r = lane; r < N; r += 64) {
            const double v = row[r];
            a1 = fma(v, w1[r], a1);
            a2 = fma(v, w2[r], a2);
        }
    }
}

// The same for TWO rows at once, sixteen 16-byte loads per lane in flight (large N: with one workgroup on the chip a row at a
// time, four loads deep, waits out an L2 round trip per KiB).  Per row the sums run in row_dot2's order: the same bits.
__device__ __forceinline__ void rows2_dot2(const double *__restrict__ rowA, const double *__restrict__ rowB, const double *w1,
                                           const double *w2, int N, int lane, double &a1A, double &a2A, double &a1B, double &a2B) {
    a1A = a2A = a1B = a2B = 0.0;
    for (int r0 = lane * 2; r0 < N; r0 += 128 * 8) {
        double2 va[8], vb[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int r = r0 + 128 * m < N ? r0 + 128 * m : lane * 2;
            va[m] = *reinterpret_cast<const double2 *>(rowA + r);
            vb[m] = *reinterpret_cast<const double2 *>(rowB + r);
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int r = r0 + 128 * m;
            if (r < N) {
                const double2 x = *reinterpret_cast<const double2 *>(w1 + r);
                const double2 y = *reinterpret_cast<const double2 *>(w2 + r);
                a1A = fma(va[m].y, x.y, fma(va[m].x, x.x, a1A));
                a2A = fma(va[m].y, y.y, fma(va[m].x, y.x, a2A));
                a1B = fma(vb[m].y, x.y, fma(vb[m].x, x.x, a1B));
                a2B = fma(vb[m].y, y.y, fma(vb[m].x, y.x, a2B));
            }
        }
    }
}

// Partial dot products of four V columns with the LDS vector w: 16 independent
// 16-byte loads per lane in flight at N = 512 (4 KiB per wave instruction group).
template <int VEC>
__device__ __forceinline__ void dot4_cols(const double *__restrict__ c0, const double *__restrict__ c1,
                                          const double *__restrict__ c2, const double *__restrict__ c3,
                                          const double *w, int N, int lane, double (&a)[4]) {
    a[0] = a[1] = a[2] = a[3] = 0.0;
    if (VEC >= 2) {
#pragma unroll 4
        for (int r = lane * 2; r < N; r += 128) {
            const double2 v0 = *reinterpret_cast<const double2 *>(c0 + r);
            const double2 v1 = *reinterpret_cast<const double2 *>(c1 + r);
            const double2 v2 = *reinterpret_cast<const double2 *>(c2 + r);
            const double2 v3 = *reinterpret_cast<const double2 *>(c3 + r);
            const double2 ww = *reinterpret_cast<const double2 *>(w + r);
            a[0] = fma(v0.y, ww.y, fma(v0.x, ww.x, a[0]));
            a[1] = fma(v1.y, ww.y, fma(v1.x, ww.x, a[1]));
            a[2] = fma(v2.y, ww.y, fma(v2.x, ww.x, a[2]));
            a[3] = fma(v3.y, ww.y, fma(v3.x, ww.x, a[3]));
        }
    } else {
#pragma unroll 4
        for (int r = lane; r < N; r += 64) {
            const double ww = w[r];
            a[0] = fma(c0[r], ww, a[0]);
            a[1] = fma(c1[r], ww, a[1]);
            a[2] = fma(c2[r], ww, a[2]);
            a[3] = fma(c3[r], ww, a[3]);
        }
    }
}

// out[col] = V[:,col] . w  for n columns.  cols == nullptr: columns 0..n-1; else the listed ones
// (stride cstep: -1 walks the bound list from the back of idx).
template <int VEC>
__device__ __forceinline__ void stream_cols(const double *__restrict__ V, int N, const int16_t *cols,
                                            int cstep, int n, const double *w, double *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave * 4; t < n; t += NW * 4) {
        int j[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int tc = (t + c < n) ? t + c : t;
            j[c] = cols ? (int)cols[tc * cstep] : tc;
        }
        double a[4];
        dot4_cols<VEC>(V + (size_t)j[0] * N, V + (size_t)j[1] * N, V + (size_t)j[2] * N, V + (size_t)j[3] * N, w, N,
                       lane, a);
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = wave_sum(a[c]);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (t + c < n) out[j[c]] = a[c];
        }
    }
}

// Up to RPW constraint rows per wavefront with ALL their loads in flight before the first use (one memory round
// trip per batch instead of one per row): row t of this wavefront is rows[t] (nullptr = none), N even, N <= 512.
// a1[t] = row . w1, a2[t] = row . w2, accumulated in the order of row_dot2 (lane's pairs r = 2*lane + 128*m).
template <int RPW>
__device__ __forceinline__ void rows_dot2_batch(const double *__restrict__ const (&rows)[RPW], const double *w1,
                                                const double *w2, int N, int lane, double (&a1)[RPW],
                                                double (&a2)[RPW]) {
    double2 g[RPW][4];
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
        const double *__restrict__ row = rows[t] ? rows[t] : rows[0];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int r = lane * 2 + 128 * m;
            g[t][m] = *reinterpret_cast<const double2 *>(row + (r < N ? r : 0));
        }
    }
    double2 x[4], y[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int r = lane * 2 + 128 * m;
        x[m] = *reinterpret_cast<const double2 *>(w1 + (r < N ? r : 0));
        y[m] = *reinterpret_cast<const double2 *>(w2 + (r < N ? r : 0));
    }
#pragma unroll
    for (int t = 0; t < RPW; ++t) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int r = lane * 2 + 128 * m;
            const double n1 = fma(g[t][m].y, x[m].y, fma(g[t][m].x, x[m].x, s1));
            const double n2 = fma(g[t][m].y, y[m].y, fma(g[t][m].x, y[m].x, s2));
            s1 = (r < N) ? n1 : s1;
            s2 = (r < N) ? n2 : s2;
        }
        a1[t] = s1;
        a2[t] = s2;
    }
}

// out = V[:, nz] * w[nz] in AXPY form: every lane owns fixed rows (the 16 bytes it loads from each 1 KiB
// slice of a column) and accumulates them in registers over the columns its wavefront takes; no cross-lane
// reduction.  Columns whose weight is exactly 0.0 are not in the list `nzl` and are never read (they would
// add exact zeros), which is what makes the gamma pass cheap on portfolio problems where most bound
// variables sit at d = 0.  The NW per-wave partial vectors are summed in wave order through `stage`
// (NW*N doubles), so the result is deterministic.  N even (16-byte loads), N <= 128*NCH.
// Extra "columns" the AXPY pass can take after the columns of V (list entries N.., see stream_matvec): the kept
// constraint rows with weights alphaL (the AB'*alphaL term of gamma, SSQP.jl:352) and q with weight 1.
struct AxpyExt {
    const double *Ct, *q, *aL;
    const int16_t *rowsE, *ra;
    int W;  // -1: no extras
};

template <int NCH, int NCOL>
__device__ __forceinline__ void stream_axpy(const double *__restrict__ V, int N, const int16_t *nzl, int nnz,
                                            const double *w, double *stage, double *out, const AxpyExt &X_,
                                            const double *addend = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double2 acc[NCH];
#pragma unroll
    for (int m = 0; m < NCH; ++m) acc[m] = make_double2(0.0, 0.0);
    for (int t = wave * NCOL; t < nnz; t += NW * NCOL) {
        const double *__restrict__ col[NCOL];
        double wj[NCOL];
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            const bool live = (t + c < nnz);
            const int j = nzl[live ? t + c : t];
            if (j < N) {
                col[c] = V + (size_t)j * N;
                wj[c] = live ? w[j] : 0.0;
            } else if (j < N + X_.W) {
                col[c] = X_.Ct + (size_t)X_.rowsE[X_.ra[j - N]] * N;
                wj[c] = live ? X_.aL[j - N] : 0.0;
            } else {
                col[c] = X_.q;
                wj[c] = live ? 1.0 : 0.0;
            }
        }
        // all NCOL*NCH loads are issued before the first use: rows beyond N are read from row 0 and dropped
        double2 v[NCH][NCOL];
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            const int r = lane * 2 + 128 * m;
            const int rr = (r < N) ? r : 0;
#pragma unroll
            for (int c = 0; c < NCOL; ++c) v[m][c] = *reinterpret_cast<const double2 *>(col[c] + rr);
        }
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            const int r = lane * 2 + 128 * m;
            const double keep = (r < N) ? 1.0 : 0.0;
#pragma unroll
            for (int c = 0; c < NCOL; ++c) {
                const double wk = wj[c] * keep;
                acc[m].x = fma(v[m][c].x, wk, acc[m].x);
                acc[m].y = fma(v[m][c].y, wk, acc[m].y);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const int r = lane * 2 + 128 * m;
        if (r < N) *reinterpret_cast<double2 *>(stage + (size_t)wave * N + r) = acc[m];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += NT) {
        double s = stage[i];
#pragma unroll
        for (int wv = 1; wv < NW; ++wv) s += stage[(size_t)wv * N + i];
        if (addend) s += addend[i];
        out[i] = s;
    }
}

// list of the indices with w[i] != 0 (all indices when `dense`), increasing; returns the count
__device__ __forceinline__ int compact_nonzero(const double *w, int N, bool dense, int16_t *list, const Lds &L);

// ------------------------------------------------------------- compaction
// pos[i] = rank of i among the free variables or -1; idx[0..K) = free indices
// (increasing, like findall, SSQP.jl:276); idx[N-1-r] = r-th bound index.
// One workgroup barrier inside: every NT-chunk is balloted first, the per-wave counts of all chunks are exchanged
// at once (ired[CNT0 + chunk*NW + wave]) and each thread derives its offsets from them.  Also scatters zB:
// zm[i] = z[i] for bound variables, 0 for free ones (SSQP.jl:286).  The caller's next barrier publishes the lists.
constexpr int CNT0 = 2 * NW + 16;  // first count slot in ired
constexpr int EVT_MAX = 16;
constexpr int HQ_CHANGED = 2 * NW + 8;  // ired slot: hq was updated by a status switch since the border columns were formed
constexpr int ROWS_DIRTY = 2 * NW + 15;  // ired slot: an inequality changed status since the row lists were formed
constexpr int SHIFT_COUNT = 2 * NW + 6;  // ired slot: status switches that updated hq / bEall by one column since their last full evaluation
constexpr int HB_DIRTY = 2 * NW + 13;  // ired slot: a bound variable with z != 0 changed status since hq was formed
template <int MPT>
__device__ __forceinline__ int compact_free(const Lds &L, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int lp[MPT];
    unsigned fm = 0;
#pragma unroll
    for (int m = 0; m < MPT; ++m) {
        const int i = m * NT + threadIdx.x;
        lp[m] = 0;
        if (m * NT < N) {  // uniform
            const bool f = (i < N) && (L.S[i < N ? i : 0] == SSQP_IN);
            const unsigned long long bm = __ballot(f);
            lp[m] = __popcll(bm & ((1ull << lane) - 1ull));
            fm |= f ? (1u << m) : 0u;
            if (lane == 0) L.ired[CNT0 + m * NW + wave] = __popcll(bm);
        }
    }
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int m = 0; m < MPT; ++m) {
        if (m * NT < N) {
            int wb = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int c = L.ired[CNT0 + m * NW + w];
                if (w < wave) wb += c;
                tot += c;
            }
            const int i = m * NT + threadIdx.x;
            if (i < N) {
                if (fm & (1u << m)) {
                    const int p = base + wb + lp[m];
                    L.pos[i] = (int16_t)p;
                    L.idx[p] = (int16_t)i;
                    L.zm[i] = 0.0;
                } else {
                    L.pos[i] = -1;
                    const int r = i - (base + wb + lp[m]);  // rank among bound variables
                    L.idx[N - 1 - r] = (int16_t)i;
                    L.zm[i] = L.z[i];
                }
            }
            base += tot;
        }
    }
    return uni(base);
}

__device__ __forceinline__ int compact_nonzero(const double *w, int N, bool dense, int16_t *list, const Lds &L) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int c0 = 0; c0 < N; c0 += NT) {
        const int i = c0 + threadIdx.x;
        const bool f = (i < N) && (dense || w[i] != 0.0);
        const unsigned long long m = __ballot(f);
        const int lp = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) L.ired[wave] = __popcll(m);
        __syncthreads();
        int wb = 0, tot = 0;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) {
            const int c = L.ired[wv];
            if (wv < wave) wb += c;
            tot += c;
        }
        if (f) list[base + wb + lp] = (int16_t)i;
        base += tot;
        __syncthreads();
    }
    return uni(base);
}

// out = V * w restricted to the columns with nonzero weight: AXPY form for even N, dot form otherwise
template <int VEC>
__device__ __forceinline__ int stream_matvec(const double *__restrict__ V, int N, const double *w, bool dense,
                                             double *stage, double *out, const Lds &L,
                                             const AxpyExt &X_ = AxpyExt{nullptr, nullptr, nullptr, nullptr, nullptr, -1}) {
    // VEC: 1 scalar loads (odd N); 2, 3, 4: 16-byte loads with N <= 512, 1024, 2048 (register slots per lane)
    if (VEC >= 2) {
        int nnz = compact_nonzero(w, N, dense, L.perm, L);
        if (X_.W >= 0) {  // append the constraint rows and q to the column list
            if ((int)threadIdx.x <= X_.W) L.perm[nnz + threadIdx.x] = (int16_t)(N + threadIdx.x);
            nnz += X_.W + 1;
            __syncthreads();
        }
        if (VEC == 2) stream_axpy<4, 4>(V, N, L.perm, nnz, w, stage, out, X_);
        else if (VEC == 3) stream_axpy<8, 2>(V, N, L.perm, nnz, w, stage, out, X_);
        else stream_axpy<16, 1>(V, N, L.perm, nnz, w, stage, out, X_);
        return nnz;
    } else {
        stream_cols<VEC>(V, N, nullptr, 1, N, w, out);
        return N;
    }
}

// ------------------------------------------------- getRowsGJr (utils.jl:49-86)
// X is W0 x nc column-major in the arena (destroyed).  Returns the number of
// kept rows; ra[0..W) their indices (increasing).  Arithmetic is the
// reference's, operation for operation (no FMA contraction), pivot = first
// maximum of |X[i, c0[j:nc]]| in c0 order.
__device__ __forceinline__ int rank_filter(double *X, int W0, int nc, double tol, const Lds &L) {
    for (int t = threadIdx.x; t < nc; t += NT) L.perm[t] = (int16_t)t;
    __syncthreads();
    int i = 0, j = 0, nrows = 0;
    while (i < W0 && j < nc) {
        KeyMin best{0.0, 0x7fffffff};  // maximise |x|  ==  minimise -|x|
        bool any = false;
        for (int t = j + (int)threadIdx.x; t < nc; t += NT) {
            const double v = -fabs(X[i + W0 * (int)L.perm[t]]);
            if (!any || v < best.v) {
                best.v = v;
                best.ord = t;
                any = true;
            }
        }
        if (!any) best.v = 1.0;  // never wins against -|x| <= 0
        best = block_keymin(best, L);
        const double m = -best.v;
        const int mj = best.ord;
        if (!(m > tol)) {  // utils.jl:61  (m <= tol, NaN falls through like Julia's findmax would not; rare)
            i += 1;
            continue;
        }
        if (threadIdx.x == 0) {
            L.ra[nrows] = (int16_t)i;
            const int16_t t = L.perm[mj];
            L.perm[mj] = L.perm[j];
            L.perm[j] = t;
        }
        nrows += 1;
        __syncthreads();
        const int n = L.perm[j];
        const double dd = X[i + W0 * n];
        __syncthreads();
        for (int t = j + (int)threadIdx.x; t < nc; t += NT) {
            const int c = L.perm[t];
            X[i + W0 * c] = X[i + W0 * c] / dd;
        }
        for (int k = threadIdx.x; k < W0; k += NT) L.dcol[k] = X[k + W0 * n];
        __syncthreads();
        const int span = nc - j;
        for (int e = threadIdx.x; e < span * W0; e += NT) {
            const int k = e % W0, t = j + e / W0;
            if (k != i) {
                const int c = L.perm[t];
                X[k + W0 * c] = sub_mul_nc(X[k + W0 * c], L.dcol[k], X[i + W0 * c]);
            }
        }
        __syncthreads();
        i += 1;
        j += 1;
    }
    __syncthreads();
    return nrows;
}

// The same filter run by ONE wavefront (no workgroup barriers): used when X is small, which is the
// common case (a handful of active rows).  Executed by wave 0 only; returns the number of kept rows.
__device__ __forceinline__ int rank_filter_wave(double *X, int W0, int nc, double tol, const Lds &L) {
    const int lane = threadIdx.x & 63;
    for (int t = lane; t < nc; t += 64) L.perm[t] = (int16_t)t;
    wave_sync();
    int i = 0, j = 0, nrows = 0;
    while (i < W0 && j < nc) {
        // first maximum of |X[i, c0[j:nc]]| in c0 order: positions t = j + lane (+64..) are visited in order, so it
        // is the lowest lane of the earliest 64-chunk that holds the maximum
        double a0 = -1.0, a1 = -1.0, a2 = -1.0;
        {
            const int t0 = j + lane, t1 = t0 + 64, t2 = t0 + 128;
            if (t0 < nc) a0 = fabs(X[i + W0 * (int)L.perm[t0]]);
            if (t1 < nc) a1 = fabs(X[i + W0 * (int)L.perm[t1]]);
            if (t2 < nc) a2 = fabs(X[i + W0 * (int)L.perm[t2]]);
        }
        const double m = wave_max_uniform(fmax(a0, fmax(a1, a2)));
        if (!(m > tol)) {
            i += 1;
            continue;
        }
        int mj;
        {
            const unsigned long long e0 = __ballot(a0 == m), e1 = __ballot(a1 == m), e2 = __ballot(a2 == m);
            mj = j + (e0 ? (__ffsll((long long)e0) - 1) : (e1 ? 64 + (__ffsll((long long)e1) - 1) : 128 + (__ffsll((long long)e2) - 1)));
        }
        if (lane == 0) {
            L.ra[nrows] = (int16_t)i;
            const int16_t t = L.perm[mj];
            L.perm[mj] = L.perm[j];
            L.perm[j] = t;
        }
        nrows += 1;
        wave_sync();
        const int n = L.perm[j];
        const double dd = X[i + W0 * n];
        for (int k = lane; k < W0; k += 64) L.dcol[k] = X[k + W0 * n];
        wave_sync();
        // Lanes over the columns c0[j:nc]; rows 0..i-1 are never read again by the filter (every later decision
        // looks at a row below), so only row i is normalised and rows i+1.. are eliminated: same values in
        // every entry that can still influence a decision as the reference's full Gauss-Jordan sweep.
        for (int t = j + lane; t < nc; t += 64) {
            const int c = L.perm[t];
            double *col = X + W0 * c;
            const double xn = col[i] / dd;
            col[i] = xn;
#pragma unroll 4
            for (int k = i + 1; k < W0; ++k) col[k] = sub_mul_nc(col[k], L.dcol[k], xn);
        }
        wave_sync();
        i += 1;
        j += 1;
    }
    return nrows;
}

// The same filter with X held in REGISTERS of one wavefront (W0 <= 8 rows, nc <= 192 columns: lane owns
// columns lane, lane+64, lane+128): no LDS traffic and no barriers inside the elimination.  The reference's
// column permutation c0 is tracked as a position per column (a swap of c0[mj] and c0[j] swaps two positions), so
// the pivot rule -- first maximum in c0 order -- and every arithmetic operation are those of utils.jl:58-83.
constexpr int RF_ROWS = 12, RF_CS = 3;
// ROWS x CS is the register tile (rows beyond W0 and columns beyond nc hold zeros / a dead position and take
// part in the arithmetic harmlessly: no guards, straight-line code).
template <int ROWS, int CS>
__device__ __forceinline__ int rank_filter_regs(const double *X, int W0, int nc, double tol, const Lds &L) {
    const int lane = threadIdx.x & 63;
    double x[CS][ROWS];
    int posn[CS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
        const int t = lane + 64 * cs;
        posn[cs] = (t < nc) ? t : 0x7fff0000;  // dead columns never qualify
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            const double v = X[(k < W0 ? k : 0) + W0 * (t < nc ? t : 0)];
            x[cs][k] = (t < nc && k < W0) ? v : 0.0;
        }
    }
    // The current row always sits in register slot 0: after a row is done (kept or purged, utils.jl:61-64) the
    // rows below move up one slot.  One compact loop body, no dynamic register index.  Columns that are no
    // longer candidates (earlier pivot columns, padding) are updated along without predication: nothing reads
    // them again.
    int i = 0, j = 0, nrows = 0;
    while (i < W0 && j < nc) {
        // this lane's best candidate in the row: largest |x|, ties -> smallest position in c0
        double am = -1.0;
        int ap = 0x7fffffff;
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
            const bool in = (posn[cs] >= j) & (posn[cs] < nc);
            const double a = in ? fabs(x[cs][0]) : -1.0;
            const bool better = (a > am) | ((a == am) & (posn[cs] < ap));  // (no short circuit: straight-line code)
            am = better ? a : am;
            ap = better ? posn[cs] : ap;
        }
        const double m = wave_max(am);  // (lanes >= nc hold -1)
        if (m > tol) {  // utils.jl:61 (uniform)
            // first maximum in c0 order: the tied lane with the smallest position (one lane but for exact ties)
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
            if (lane == 0) L.ra[nrows] = (int16_t)i;
            nrows += 1;
            // c0[mj] <-> c0[j]; then the pivot column (now at position j) broadcasts its entries of this row and below
            double dcol[ROWS];
            if (CS == 1) {  // one column per lane: the pivot column is the winning lane's
                const int pz = posn[0];
                posn[0] = (pz == mpos) ? j : ((pz == j) ? mpos : pz);
#pragma unroll
                for (int k = 0; k < ROWS; ++k) dcol[k] = readlane_f64(x[0][k], mlane);
            } else {
#pragma unroll
            for (int k = 0; k < ROWS; ++k) dcol[k] = 0.0;
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) {
                const int pz = posn[cs];
                posn[cs] = (pz == mpos) ? j : ((pz == j) ? mpos : pz);
                const unsigned long long own = __ballot(posn[cs] == j);
                if (own) {  // uniform: the owner lane holds the pivot column in this slot
                    const int src = __ffsll((long long)own) - 1;
#pragma unroll
                    for (int k = 0; k < ROWS; ++k) dcol[k] = readlane_f64(x[cs][k], src);
                }
            }
            }
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) {
                const double xn = x[cs][0] / dcol[0];  // utils.jl:68-70 (IEEE division, like the reference)
#pragma unroll
                for (int k = 1; k < ROWS; ++k)  // utils.jl:71-78, rows below (rows above are never read again)
                    x[cs][k] = sub_mul_nc(x[cs][k], dcol[k], xn);
            }
            j += 1;
        }
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
#pragma unroll
            for (int k = 0; k + 1 < ROWS; ++k) x[cs][k] = x[cs][k + 1];
            x[cs][ROWS - 1] = 0.0;
        }
        i += 1;
    }
    wave_sync();
    return nrows;
}

// --------------------------------------------------- bordered LDL' in the arena
// fac: packed lower triangle (column j holds rows j..R-1) of the R x R matrix
//     [ V_FF  .   . ]      R = K + W + 1
//     [ AE    0   . ]
//     [ c'    0   0 ]
// bordered_ldl eliminates columns jb..je-1 (right-looking, one barrier per column) and applies every
// update to ALL later columns, border included.  After columns 0..K-1: column j holds the unscaled
// entries a(i,j), rd[j] = 1/d_j, unit-lower L(i,j) = a(i,j)*rd[j]; the border rows of those columns are
// L^-1 [AE' c], and the trailing (W+1) x (W+1) block is the Schur complement -[AE;c'] V_FF^-1 [AE' c].
// Returns false when a pivot is not > 0 (where the reference's cholesky throws PosDefException).
__device__ __forceinline__ bool bordered_ldl1(double *fac, double *rd, int jb, int je, int R, const Lds &L) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool ok = true;
    for (int j = jb; j < je; ++j) {
        __syncthreads();
        const int oj = coloff(j, R);
        const double d = fac[oj];
        if (!(d > 0.0)) {
            ok = false;
            break;
        }
        const double r = fast_rcp(d);
        if (threadIdx.x == 0) rd[j] = r;
        // four columns per wavefront trip: the column-j entry of a row is read once and feeds four
        // independent read-modify-writes, so their LDS latencies overlap
        for (int k0 = j + 1 + wave * 4; k0 < R; k0 += NW * 4) {
            double f[4];
            int oc[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = k0 + c;
                const bool live = k < R;
                f[c] = live ? fac[oj + k - j] * r : 0.0;
                oc[c] = coloff(live ? k : k0, R) - (live ? k : k0);  // fac[oc + i] is element (i, k)
            }
            for (int i = k0 + lane; i < R; i += 64) {
                const double aij = fac[oj + i - j];
                double v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = (i >= k0 + c && k0 + c < R) ? fac[oc[c] + i] : 0.0;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (i >= k0 + c && k0 + c < R) fac[oc[c] + i] = fma(-f[c], aij, v[c]);
            }
        }
    }
    __syncthreads();
    return ok;
}

// Panel version: four columns per step.  Wavefront 0 factors the 4-wide panel (rows j..R-1) in registers --
// lane = row, the rows of the 4x4 diagonal block are broadcast with v_readlane -- then all wavefronts apply
// the rank-4 update to the trailing columns.  Two workgroup barriers per FOUR columns instead of one per
// column.  Needs R - jb <= 256 (four row slots per lane); otherwise the one-column version runs.

// Blocked right-looking version for ANY size (used above 256 rows, where the factor lives in the workgroup's global
// arena): PB = 16 columns per panel.
//   1. one wavefront factors the PB x PB diagonal block in registers (lane = row, v_readlane broadcasts) and leaves the
//      multipliers m(c2,c) = a(j0+c2, j0+c) / d_c and the reciprocal pivots in LDS;
//   2. every thread applies those eliminations to its rows below the block (rows are independent: 16 entries in
//      registers, 120 multiply-adds, one coalesced load/store per column);
//   3. the trailing matrix (border rows and Schur block included) gets the rank-PB update on the matrix cores: each
//      16 x 16 tile of C is loaded ONCE, takes four chained v_mfma_f64_16x16x4_f64 (one per 4 panel columns) and is
//      stored once -- four times the arithmetic intensity of the 4-column panel version.
// Same storage convention as bordered_ldl1 / bordered_ldl: column j holds the unscaled a(i,j), rd[j] = 1/d_j.
__device__ __forceinline__ bool bordered_ldl_blocked(double *fac, double *rd, int jb, int je, int R, const Lds &L) {
    constexpr int PB = 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *mult = L.arena;             // PB x PB multipliers, column-major: mult[c2 + PB*c] (c2 > c)
    double *prd = L.arena + PB * PB;    // PB reciprocal pivots
    for (int j0 = jb; j0 < je; j0 += PB) {
        const int nb = (je - j0 < PB) ? je - j0 : PB;
        __syncthreads();
        if (wave == 0) {  // ---- diagonal block: rows j0 .. j0+nb-1 (lane = row of the block)
            double a[PB];
#pragma unroll
            for (int c = 0; c < PB; ++c) {
                const bool live = (c < nb) && (lane < nb) && (lane >= c);
                const int jc = j0 + (c < nb ? c : nb - 1);
                const double v = fac[coloff(jc, R) + (live ? j0 + lane - jc : 0)];
                a[c] = live ? v : 0.0;
            }
            bool ok = true;
#pragma unroll
            for (int c = 0; c < PB; ++c) {
                if (c < nb) {  // uniform
                    const double d = readlane_f64(a[c], c);
                    if (!(d > 0.0)) ok = false;
                    const double r = fast_rcp(d);
                    if (lane == 0) prd[c] = r;
#pragma unroll
                    for (int c2 = c + 1; c2 < PB; ++c2) {
                        if (c2 < nb) {
                            const double m = readlane_f64(a[c], c2) * r;  // a(j0+c2, j0+c) / d_c
                            if (lane == 0) mult[c2 + PB * c] = m;
                            a[c2] = (lane >= c2) ? fma(-m, a[c], a[c2]) : a[c2];
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 1; c < PB; ++c) {
                const int jc = j0 + c;
                if (c < nb && lane < nb && lane >= c) fac[coloff(jc, R) + j0 + lane - jc] = a[c];
            }
            if (lane == 0) {
                L.ired[2 * NW + 5] = ok ? 1 : 0;
#pragma unroll
                for (int c = 0; c < PB; ++c)
                    if (c < nb) rd[j0 + c] = prd[c];
            }
        }
        __syncthreads();
        if (!L.ired[2 * NW + 5]) {
            __syncthreads();
            return false;
        }
        // ---- rows below the diagonal block: the same eliminations, row by row
        for (int i = j0 + nb + tid; i < R; i += NT) {
            double a[PB];
#pragma unroll
            for (int c = 0; c < PB; ++c) {
                const int jc = j0 + (c < nb ? c : nb - 1);
                a[c] = fac[coloff(jc, R) + i - jc];
            }
#pragma unroll
            for (int c = 0; c < PB; ++c) {
                if (c < nb) {
#pragma unroll
                    for (int c2 = c + 1; c2 < PB; ++c2)
                        if (c2 < nb) a[c2] = fma(-mult[c2 + PB * c], a[c], a[c2]);
                }
            }
#pragma unroll
            for (int c = 1; c < PB; ++c)
                if (c < nb) fac[coloff(j0 + c, R) + i - (j0 + c)] = a[c];
        }
        __syncthreads();
        // ---- rank-nb update of the trailing columns: D = C - A B, A = a(rows, panel), B = (a(cols, panel) / d)'
        {
            using f64x4 = __attribute__((ext_vector_type(4))) double;
            const int jt = j0 + nb;                      // first trailing column
            const int nt = (R - jt + 15) >> 4;           // tiles per dimension
            const int kk = lane >> 4, l15 = lane & 15;
            const int ntile = nt * (nt + 1) / 2;
            for (int t = wave; t < ntile; t += NW) {
                int tk = 0, f = t;  // lower-triangular tile index -> (ti >= tk), column-major over tile columns
                while (f >= nt - tk) {
                    f -= nt - tk;
                    ++tk;
                }
                const int ti = tk + f;
                const int i0 = jt + 16 * ti, k0 = jt + 16 * tk;
                const int ra_ = i0 + l15, cb_ = k0 + l15;
                const int colc = (cb_ < R) ? cb_ : jt;
                const int occ = coloff(colc, R) - colc;
                f64x4 cv;
                bool ok[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = i0 + kk + 4 * g;
                    ok[g] = (cb_ < R) && (row < R) && (row >= cb_);
                    cv[g] = fac[occ + (ok[g] ? row : colc)];
                    cv[g] = ok[g] ? cv[g] : 0.0;
                }
#pragma unroll
                for (int kb = 0; kb < PB / 4; ++kb) {
                    const int pc = 4 * kb + kk;          // this lane's panel column in this k-block
                    const bool live = pc < nb;
                    const int jc = j0 + (live ? pc : 0);
                    const int opc = coloff(jc, R) - jc;
                    const double av = fac[opc + (ra_ < R ? ra_ : jc)];
                    const double bv = fac[opc + (cb_ < R ? cb_ : jc)];
                    const double a = (ra_ < R && live) ? -av : 0.0;
                    const double bb = (cb_ < R && live) ? bv * rd[jc] : 0.0;
                    if (4 * kb < nb) cv = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, cv, 0, 0, 0);  // uniform
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = i0 + kk + 4 * g;
                    if (ok[g]) fac[occ + row] = cv[g];
                }
            }
        }
    }
    __syncthreads();
    return true;
}

__device__ __forceinline__ bool bordered_ldl(double *fac, double *rd, int jb, int je, int R, const Lds &L) {
    constexpr int RS = 4;
    // above four register slots of rows: the blocked version (the factor is in the global arena there, and the LDS
    // arena is free for the panel's multipliers)
    if (R - jb > 64 * RS) return (fac != L.arena) ? bordered_ldl_blocked(fac, rd, jb, je, R, L) : bordered_ldl1(fac, rd, jb, je, R, L);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    SUBPHASE_DECL(tsub);
    for (int j = jb; j < je; j += 4) {
        const int nb = (je - j < 4) ? je - j : 4;
        __syncthreads();
        SUBPHASE(15, tsub);
        if (wave == 0) {  // ---- panel factorisation in registers
            // (loads are unconditional from clamped, always valid addresses; a branch around an LDS load
            //  would serialise the loads -- each behind its own s_waitcnt)
            double a[RS][4];
            int oc[4];
            const int nsl = (R - j + 63) >> 6;  // row slots in use (uniform)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int jc = j + (c < nb ? c : nb - 1);
                oc[c] = coloff(jc, R) - jc;
            }
#pragma unroll
            for (int m = 0; m < RS; ++m) {
                const int i = j + lane + 64 * m;
#pragma unroll
                for (int c = 0; c < 4; ++c) a[m][c] = 0.0;
                if (m < nsl)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int jc = j + (c < nb ? c : nb - 1);
                    const bool live = (c < nb) && (i < R) && (i >= jc);
                    const double v = fac[oc[c] + (live ? i : jc)];
                    a[m][c] = live ? v : 0.0;
                }
            }
            bool ok = true;
            double rc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < nb) {
                    const double d = readlane_f64(a[0][c], c);  // a(j+c, j+c)
                    if (!(d > 0.0)) ok = false;
                    const double r = fast_rcp(d);
                    rc[c] = r;
#pragma unroll
                    for (int c2 = c + 1; c2 < 4; ++c2) {
                        if (c2 < nb) {
                            const double l = readlane_f64(a[0][c], c2) * r;  // a(j+c2, j+c) / d
#pragma unroll
                            for (int m = 0; m < RS; ++m) {
                                if (m < nsl) {
                                    const int i = j + lane + 64 * m;
                                    const double t = fma(-l, a[m][c], a[m][c2]);
                                    a[m][c2] = (i >= j + c2) ? t : a[m][c2];
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < RS; ++m) {
                const int i = j + lane + 64 * m;
#pragma unroll
                for (int c = 1; c < 4; ++c)
                    if (c < nb && i < R && i >= j + c) fac[oc[c] + i] = a[m][c];
            }
            if (lane == 0) {
                L.ired[2 * NW + 5] = ok ? 1 : 0;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < nb) rd[j + c] = rc[c];
            }
        }
        SUBPHASE(13, tsub);
        __syncthreads();
        if (!L.ired[2 * NW + 5]) {
            __syncthreads();
            return false;
        }
        // ---- rank-nb update of the trailing columns (border and Schur block included) on the matrix cores:
        // the trailing triangle is cut into 16x16 tiles and each tile gets ONE v_mfma_f64_16x16x4_f64,
        //   D(16x16) = C - A(16x4) * B(4x16),  A = L(rows, panel),  B = (L(cols, panel) * 1/d)'
        // f64 MFMA fragment layout (cdna_hip_programming.md section 3): lane l holds A[l&15][l>>4] and B[l>>4][l&15];
        // C/D register g of lane l is element (row (l>>4) + 4g, col l&15).
        {
            using f64x4 = __attribute__((ext_vector_type(4))) double;
            const int jt = j + nb;                       // first trailing column
            const int nt = (R - jt + 15) >> 4;           // tiles per dimension
            const int kk = lane >> 4, l15 = lane & 15;
            const int jc = j + (kk < nb ? kk : nb - 1);  // this lane's panel column
            const int opc = coloff(jc, R) - jc;
            const double rpk = (kk < nb) ? rd[jc] : 0.0;
            for (int t = wave; t < nt * (nt + 1) / 2; t += NW) {
                int tk = 0, f = t;  // lower-triangular tile index -> (ti >= tk), column-major over tile columns
                while (f >= nt - tk) {
                    f -= nt - tk;
                    ++tk;
                }
                const int ti = tk + f;
                const int i0 = jt + 16 * ti, k0 = jt + 16 * tk;
                const int ra_ = i0 + l15, cb_ = k0 + l15;
                const double av = fac[opc + (ra_ < R ? ra_ : jc)];
                const double bv = fac[opc + (cb_ < R ? cb_ : jc)];
                const double a = (ra_ < R && kk < nb) ? -av : 0.0;
                const double bb = (cb_ < R && kk < nb) ? bv * rpk : 0.0;
                const int colc = (cb_ < R) ? cb_ : jt;
                const int occ = coloff(colc, R) - colc;
                f64x4 cv;
                bool ok[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = i0 + kk + 4 * g;
                    ok[g] = (cb_ < R) && (row < R) && (row >= cb_);
                    cv[g] = fac[occ + (ok[g] ? row : colc)];
                    cv[g] = ok[g] ? cv[g] : 0.0;
                }
                cv = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, cv, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = i0 + kk + 4 * g;
                    if (ok[g]) fac[occ + row] = cv[g];
                }
            }
        }
        SUBPHASE(14, tsub);
    }
    __syncthreads();
    return true;
}

// Solve the unit upper system L' x = v in place for the n columns j0..j0+n-1 (v[0..n)), one wavefront.
__device__ __forceinline__ void back_substitute(const double *fac, const double *rd, double *v, int j0, int n,
                                                int R) {
    const int lane = threadIdx.x & 63;
    for (int j = n - 1; j > 0; --j) {
        const double xj = v[j];
        for (int i = lane; i < j; i += 64) {
            const double lij = fac[coloff(j0 + i, R) + j - i] * rd[j0 + i];
            v[i] = fma(-lij, xj, v[i]);
        }
        wave_sync();
    }
}


// =====================================================================================================
// Incremental KKT engine.  The reference refactors V[F,F] from scratch on every pass (SSQP.jl:322) although F
// changes by one index at a time.  Here the unit-lower LDL' factor of V[F,F] is
// KEPT in LDS across passes, with the free variables in insertion order:
//   release of a variable  -> append one row  (one forward substitution)
//   blocked variable(s)    -> delete a row/column (rank-1 update of the trailing block + compaction)
// and each pass only does: forward substitution of the border [AE' c], the (W+1)^2 Schur block, lambda, one
// back substitution.  Same KKT system as the from-scratch path; the pivot order differs, results agree to
// rounding.  Column c of the factor lives at cofs(c) = c*RC - c(c-1)/2 with rows c..RC-1 (fixed capacity RC,
// so appending a row moves nothing): slot 0 is d_c, the rest L(r,c).  Rows are spread over lanes: row r is
// lane r&63 of register slot r>>6 (two slots: K <= 128).
// =====================================================================================================
constexpr int INC_RBM = 12;   // border right-hand sides the engine carries (W + 1 <= 12)
constexpr int INC_KMAX = 256; // up to four register slots of 64 rows

struct Inc {
    double *fcol, *rdv, *Y;  // factor, reciprocal pivots, border right-hand sides (RC per column)
    int16_t *ord, *fpos;     // row -> variable, variable -> row (or -1)
    int RC;
};
__device__ __forceinline__ int cofs(int c, int RC) { return c * RC - ((c * (c - 1)) >> 1); }

// value of row r (uniform) from the register slots
template <int SL>
__device__ __forceinline__ double row_bcast(const double (&sl)[SL], int r) {
    if (SL == 1) return readlane_f64(sl[0], r);
    const int t = r >> 6, l = r & 63;
    if (t == 0) return readlane_f64(sl[0], l);
    if (SL == 2 || t == 1) return readlane_f64(sl[1], l);
    if (t == 2) return readlane_f64(sl[SL > 2 ? 2 : 0], l);
    return readlane_f64(sl[SL > 3 ? 3 : 0], l);
}

// SL = register slots in use: 1 when K <= 64 (the common case), 2 up to 128, 4 up to 256.  The factor pointers
// may be LDS (K up to RC rows fit the arena) or the workgroup's global arena (larger K); the code is the same.

// Four columns c0..c0+3 of the kept factor for this lane's rows (entries on or above the diagonal and beyond row K
// read as 0, so a step can run unpredicated): all loads issued together; callers prefetch the next block while
// they work on the current one, so one LDS latency is paid per four elimination steps instead of per step.
template <int SL>
__device__ __forceinline__ void inc_load_cols4(const Inc &I, int K, int c0, double (&l)[4][SL]) {
    const int lane = threadIdx.x & 63;
    const int RC = uni(I.RC);
    K = uni(K);
    c0 = uni(c0);
    const int kl = (K > 0) ? K - 1 : 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = (c0 + u < K) ? c0 + u : kl;  // scalar: column offsets stay on the scalar unit
        const int oc = cofs(c, RC) - c;
        const int lim = (c0 + u < K) ? c : 0x7fffffff;
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int r = lane + 64 * t;
            const int rk = (r < K) ? r : -1;
            const int rm = (r < RC) ? r : RC - 1;
            const double v = I.fcol[oc + (rm > c ? rm : c)];  // (always inside column c)
            l[u][t] = (rk > lim) ? v : 0.0;
        }
    }
}

// Append variable j as row K (one wavefront).  Returns false when the new pivot is not > 0.
template <int SL>
__device__ __forceinline__ bool inc_append(const Inc &I, int K, int j, const double *__restrict__ V, int N) {
    const int lane = threadIdx.x & 63;
    const int RC = uni(I.RC);
    K = uni(K);
    const double *__restrict__ col = V + (size_t)j * N;
    const double vjj = col[j];
    double y[SL];
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + 64 * t;
        y[t] = (r < K) ? col[I.ord[r < K ? r : 0]] : 0.0;
    }
    {
        double lc[4][SL], ln[4][SL];
        inc_load_cols4<SL>(I, K, 0, lc);
        for (int c0 = 0; c0 < K; c0 += 4) {
            inc_load_cols4<SL>(I, K, c0 + 4, ln);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (c0 + u < K) {  // uniform
                    const double yc = row_bcast<SL>(y, c0 + u);
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
    double part = 0.0;
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + 64 * t;
        const double tr = (r < K) ? y[t] * I.rdv[r < K ? r : 0] : 0.0;
        part = fma(y[t], tr, part);
        if (r < K) I.fcol[cofs(r, RC) + K - r] = tr;
    }
    const double dnew = vjj - wave_sum(part);
    if (lane == 0) {
        I.fcol[cofs(K, RC)] = dnew;
        I.rdv[K] = fast_rcp(dnew);
        I.ord[K] = (int16_t)j;
        I.fpos[j] = (int16_t)K;
    }
    wave_sync();
    return dnew > 0.0;
}

// Rank-1 part of deleting row p: L33 D3 L33' += d_p l l'  (one wavefront).
template <int SL>
__device__ __forceinline__ void inc_delete_update(const Inc &I, int K, int p) {
    const int lane = threadIdx.x & 63;
    const int RC = uni(I.RC);
    K = uni(K);
    p = uni(p);
    const int op = cofs(p, RC) - p;
    double d[SL], w[SL];
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + 64 * t;
        d[t] = (r < K) ? I.fcol[cofs(r < K ? r : 0, RC)] : 1.0;
        w[t] = (r > p && r < K) ? I.fcol[op + r] : 0.0;
    }
    double alpha = row_bcast<SL>(d, p);
    for (int k = p + 1; k < K; ++k) {
        const int ok = cofs(k, RC) - k;
        double l[SL];
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int r = lane + 64 * t;
            l[t] = I.fcol[ok + ((r > k && r < K) ? r : k)];
        }
        const double pk = row_bcast<SL>(w, k);
        const double dk = row_bcast<SL>(d, k);
        const double dn = fma(alpha * pk, pk, dk);
        const double rdn = fast_rcp(dn);
        const double beta = pk * alpha * rdn;
        alpha = alpha * dk * rdn;
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int r = lane + 64 * t;
            if (r == k) d[t] = dn;
            if (r > k && r < K) {
                w[t] = fma(-pk, l[t], w[t]);
                I.fcol[ok + r] = fma(beta, w[t], l[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + 64 * t;
        if (r > p && r < K) {
            I.fcol[cofs(r, RC)] = d[t];
            I.rdv[r] = fast_rcp(d[t]);
        }
    }
    wave_sync();
}

// Physically remove row/column p (one wavefront, no workgroup barrier).
template <int SL>
__device__ __forceinline__ void inc_delete_compact(const Inc &I, int K, int p) {
    const int lane = threadIdx.x & 63;
    const int RC = uni(I.RC);
    K = uni(K);
    p = uni(p);
    for (int c = 0; c < p; ++c) {  // (A) columns c < p lose row p: rows r > p move up by one inside the column
        const int oc = cofs(c, RC) - c;
        double a[SL];
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int r = lane + 64 * t;
            a[t] = (r > p && r < K) ? I.fcol[oc + r] : 0.0;
        }
        wave_sync();
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int r = lane + 64 * t;
            if (r > p && r < K) I.fcol[oc + r - 1] = a[t];
        }
    }
    {  // bookkeeping of rows > p: read everything, then write
        double qv[SL];
        int vv[SL];
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int r = lane + 64 * t;
            const bool m = r > p && r < K;
            qv[t] = m ? I.rdv[r] : 0.0;
            vv[t] = m ? I.ord[r] : 0;
        }
        const int vdel = I.ord[p];
        wave_sync();
        if (lane == 0) I.fpos[vdel] = -1;
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int r = lane + 64 * t;
            if (r > p && r < K) {
                I.rdv[r - 1] = qv[t];
                I.ord[r - 1] = (int16_t)vv[t];
                I.fpos[vv[t]] = (int16_t)(r - 1);
            }
        }
    }
    for (int c = p + 1; c < K; ++c) {  // (B) column c -> c-1, ascending: the target was vacated one step earlier
        const int oc = cofs(c, RC) - c, on = cofs(c - 1, RC) - (c - 1);
        double a[SL];
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int r = lane + 64 * t;
            a[t] = (r >= c && r < K) ? I.fcol[oc + r] : 0.0;
        }
        wave_sync();
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int r = lane + 64 * t;
            if (r >= c && r < K) I.fcol[on + r - 1] = a[t];
        }
    }
    wave_sync();
}

// Forward substitution L y = b for the border right-hand sides, in place in I.Y.  Right-hand sides are taken NR
// at a time by the wavefronts (k, k+NW, ... by wavefront k & (NW-1)): no exchange between wavefronts; with up to
// 12 right-hand sides NR = 3 finishes them in one sweep over the factor.
template <int SL, int NR>
__device__ __forceinline__ void inc_forward_border(const Inc &I, int K, int nrhs, const int16_t *phys) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int RC = uni(I.RC);
    K = uni(K);
    nrhs = uni(nrhs);
    for (int b0 = wave; b0 < nrhs; b0 += NR * NW) {
        double *Yp[NR];
        double y[NR][SL];
        int ph[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int b = b0 + NW * q;
            ph[q] = phys ? (int)phys[b < nrhs ? b : b0] : (b < nrhs ? b : b0);
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            Yp[q] = I.Y + (size_t)uni(ph[q]) * RC;
#pragma unroll
            for (int t = 0; t < SL; ++t) {
                const int r = lane + 64 * t;
                y[q][t] = (r < K) ? Yp[q][r] : 0.0;
            }
        }
        {
            double lc[4][SL], ln[4][SL];
            inc_load_cols4<SL>(I, K, 0, lc);
            for (int c0 = 0; c0 < K; c0 += 4) {
                inc_load_cols4<SL>(I, K, c0 + 4, ln);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (c0 + u < K) {  // uniform
                        double cq[NR];
#pragma unroll
                        for (int q = 0; q < NR; ++q) cq[q] = row_bcast<SL>(y[q], c0 + u);
#pragma unroll
                        for (int q = 0; q < NR; ++q)
#pragma unroll
                            for (int t = 0; t < SL; ++t) y[q][t] = fma(-lc[u][t], cq[q], y[q][t]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < SL; ++t) lc[u][t] = ln[u][t];
            }
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int b = b0 + NW * q;
            if (b < nrhs) {  // uniform per wavefront
#pragma unroll
                for (int t = 0; t < SL; ++t) {
                    const int r = lane + 64 * t;
                    if (r < K) Yp[q][r] = y[q][t];
                }
            }
        }
    }
}

// Rows r0..K-1 of the forward substitution for the border columns 0..nslot-1 of I.Y (their raw right-hand sides
// already stored in those rows), given that rows < r0 are done: y_r = b_r - sum_{c<r} L(r,c) y_c, lanes over c.
// After an append only the new last row is missing: one gathered row of L and one wavefront sum per column
// instead of a sweep over the whole factor.  Columns over the wavefronts (k, k+NW, k+2NW by wavefront k).
template <int SL>
__device__ __forceinline__ void inc_forward_rows(const Inc &I, int K, int r0, int nslot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int RC = uni(I.RC);
    K = uni(K);
    r0 = uni(r0);
    nslot = uni(nslot);
    int o[SL];
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int c = lane + 64 * t;
        o[t] = cofs(c < K ? c : 0, RC) - (c < K ? c : 0);
    }
    for (int r = r0; r < K; ++r) {
        double lr[SL];
#pragma unroll
        for (int t = 0; t < SL; ++t) {
            const int c = lane + 64 * t;
            const double v = I.fcol[o[t] + (c < r ? r : (c < K ? c : 0))];  // L(r, c)
            lr[t] = (c < r) ? v : 0.0;
        }
        for (int s0 = wave; s0 < nslot; s0 += 3 * NW) {
            double part[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int sl = s0 + NW * q;
                const double *Ys = I.Y + (size_t)(sl < nslot ? sl : s0) * RC;
                double a = 0.0;
#pragma unroll
                for (int t = 0; t < SL; ++t) {
                    const int c = lane + 64 * t;
                    const double yv = Ys[c < K ? c : 0];
                    a = fma(lr[t], (c < r) ? yv : 0.0, a);  // (rows >= r hold raw right-hand sides: masked)
                }
                part[q] = a;
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int sl = s0 + NW * q;
                if (sl < nslot) {  // uniform per wavefront
                    const double sum = wave_sum(part[q]);
                    double *Ys = I.Y + (size_t)sl * RC;
                    if (lane == 0) Ys[r] = Ys[r] - sum;
                }
            }
        }
        wave_sync();  // row r of this wavefront's columns feeds its row r+1
    }
}

// Back substitution L' x = v (unit upper), v in the register slots, one wavefront.
template <int SL>
__device__ __forceinline__ void inc_backward(const Inc &I, int K, double (&v)[SL]) {
    const int lane = threadIdx.x & 63;
    const int RC = uni(I.RC);
    K = uni(K);
    int o[SL];
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int c = lane + 64 * t;
        o[t] = cofs(c < K ? c : 0, RC) - (c < K ? c : 0);
    }
    // rows K-1 .. 1 in blocks of four, the next block's entries L(r, c) in flight while this one is applied
    auto load4 = [&](int r0, double (&l)[4][SL]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 - u;
#pragma unroll
            for (int t = 0; t < SL; ++t) {
                const int c = lane + 64 * t;
                const bool live = (r > 0) && (c < r);
                const double x = I.fcol[o[t] + (live ? r : (c < K ? c : 0))];  // L(r, c): row r of column c
                l[u][t] = live ? x : 0.0;
            }
        }
    };
    double lc[4][SL], ln[4][SL];
    load4(K - 1, lc);
    for (int r0 = K - 1; r0 > 0; r0 -= 4) {
        load4(r0 - 4, ln);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r0 - u > 0) {  // uniform
                const double xr = row_bcast<SL>(v, r0 - u);
#pragma unroll
                for (int t = 0; t < SL; ++t) v[t] = fma(-lc[u][t], xr, v[t]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < SL; ++t) lc[u][t] = ln[u][t];
    }
}

// v = D^-1 (Y_A alphaL + y_c), alpha = -L'^-1 v, scattered to gam (one wavefront)
template <int SL>
__device__ __forceinline__ void inc_alpha(const Inc &I, int K, int W, int W0, const int16_t *phys, const double *aL,
                                          double *gam) {
    const int lane = threadIdx.x & 63;
    K = uni(K);
    W = uni(W);
    W0 = uni(W0);
    const double *Yc = I.Y + (size_t)W0 * I.RC;
    double v[SL];
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + 64 * t;
        v[t] = (r < K) ? Yc[r] : 0.0;
    }
    // lane w fetches the slot and the weight of border column w once; they are handed out by v_readlane below
    const int physv = phys[lane < W ? lane : 0];
    const double aLv = aL[lane < W ? lane : 0];
    for (int w0 = 0; w0 < W; w0 += 4) {  // four border columns per trip, their loads issued together
        double yw[4][SL], aw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int w = (w0 + u < W) ? w0 + u : w0;
            const double *Yw = I.Y + (size_t)__builtin_amdgcn_readlane(physv, w) * I.RC;
            aw[u] = readlane_f64(aLv, w);
#pragma unroll
            for (int t = 0; t < SL; ++t) {
                const int r = lane + 64 * t;
                yw[u][t] = Yw[r < K ? r : 0];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (w0 + u < W) {  // uniform
#pragma unroll
                for (int t = 0; t < SL; ++t) {
                    const int r = lane + 64 * t;
                    v[t] = (r < K) ? fma(yw[u][t], aw[u], v[t]) : v[t];
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + 64 * t;
        v[t] *= (r < K) ? I.rdv[r] : 0.0;
    }
    inc_backward<SL>(I, K, v);
#pragma unroll
    for (int t = 0; t < SL; ++t) {
        const int r = lane + 64 * t;
        if (r < K) gam[I.ord[r]] = -v[t];
    }
}

// dispatch on the number of register slots the current K needs
#define INC_BY_SLOTS(K_, CALL1, CALL2, CALL4) \
    do {                                      \
        if ((K_) <= 64) { CALL1; }            \
        else if ((K_) <= 128) { CALL2; }      \
        else { CALL4; }                       \
    } while (0)

// factor sync by ONE wavefront: delete the rows whose variable left F (highest row first), append the new free
// variables by increasing index.  Returns the new row count, or -1 when an appended pivot is not > 0.
__device__ __forceinline__ int inc_sync_w0(const Inc &I, int Kf, const Lds &L, const double *__restrict__ V, int N,
                                           int K, int &nAppended, int &minDel) {
    const int lane = threadIdx.x & 63;
    nAppended = 0;
    minDel = 0x7fffffff;  // lowest deleted row: the rows below it keep their forward-substituted border entries
    if (Kf < 0) {
        minDel = 0;
        for (int i = lane; i < N; i += 64) L.fpos[i] = -1;
        Kf = 0;
        wave_sync();
    }
    SUBPHASE_DECL(tscan);
    for (int t = (Kf - 1) >> 6; t >= 0; --t) {  // row chunks from the top: deleting row p leaves the rows below p in place
        const int r = lane + 64 * t;
        const bool dead = (r < Kf) && (L.S[I.ord[r < Kf ? r : 0]] != SSQP_IN);
        unsigned long long dm = __ballot(dead);
        while (dm) {
            const int pdel = 64 * t + 63 - __clzll(dm);
            dm &= ~(1ull << (pdel - 64 * t));
            SUBPHASE_DECL(tdel);
            INC_BY_SLOTS(Kf, inc_delete_update<1>(I, Kf, pdel), inc_delete_update<2>(I, Kf, pdel),
                         inc_delete_update<4>(I, Kf, pdel));
            SUBPHASE(16, tdel);
            INC_BY_SLOTS(Kf, inc_delete_compact<1>(I, Kf, pdel), inc_delete_compact<2>(I, Kf, pdel),
                         inc_delete_compact<4>(I, Kf, pdel));
            SUBPHASE(17, tdel);
            PCOUNT(24);
            minDel = pdel < minDel ? pdel : minDel;
            Kf -= 1;
        }
    }
    SUBPHASE(21, tscan);
    // new free variables: the entries of the free list idx[0..K) (increasing index) that have no row yet
    for (int c0 = 0; c0 < K; c0 += 64) {
        const int k = c0 + lane;
        const int iv = L.idx[k < K ? k : 0];
        const bool f = (k < K) && (L.fpos[iv] < 0);
        unsigned long long m = __ballot(f);
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int jn = __builtin_amdgcn_readlane(iv, b);
            bool ok1 = true;
            SUBPHASE_DECL(tapp);
            INC_BY_SLOTS(Kf, ok1 = inc_append<1>(I, Kf, jn, V, N), ok1 = inc_append<2>(I, Kf, jn, V, N),
                         ok1 = inc_append<4>(I, Kf, jn, V, N));
            SUBPHASE(18, tapp);
            PCOUNT(25);
            if (!ok1) return -1;
            Kf += 1;
            nAppended += 1;
        }
    }
    SUBPHASE(22, tscan);
    return Kf;
}

// hB partial vector of one of the two streaming wavefronts (which = 0/1): the nonzero-weight columns are found
// on the fly (no list, no barrier); this wavefront takes every second one, four at a time.
template <int NCH>
__device__ __forceinline__ int hb_partial(const double *__restrict__ V, int N, const double *w, int which,
                                          double *stage) {
    const int lane = threadIdx.x & 63;
    double2 acc[NCH];
#pragma unroll
    for (int m = 0; m < NCH; ++m) acc[m] = make_double2(0.0, 0.0);
    int cj[4] = {0, 0, 0, 0};
    int have = 0, seen = 0;
    auto flush = [&](int n) {
        const double *__restrict__ col[4];
        double wj[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = (c < n) ? cj[c] : cj[0];
            col[c] = V + (size_t)j * N;
            wj[c] = (c < n) ? w[j] : 0.0;
        }
        double2 v[NCH][4];
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            const int r = lane * 2 + 128 * m;
            const int rr = (r < N) ? r : 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[m][c] = *reinterpret_cast<const double2 *>(col[c] + rr);
        }
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            const int r = lane * 2 + 128 * m;
            const double keep = (r < N) ? 1.0 : 0.0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double wk = wj[c] * keep;
                acc[m].x = fma(v[m][c].x, wk, acc[m].x);
                acc[m].y = fma(v[m][c].y, wk, acc[m].y);
            }
        }
    };
    for (int c0 = 0; c0 < N; c0 += 64) {
        const int i = c0 + lane;
        unsigned long long m = __ballot((i < N) && (w[i < N ? i : 0] != 0.0));
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            if ((seen & 1) == which) {
                if (have == 0) cj[0] = c0 + b;
                else if (have == 1) cj[1] = c0 + b;
                else if (have == 2) cj[2] = c0 + b;
                else cj[3] = c0 + b;
                have += 1;
                if (have == 4) {
                    flush(4);
                    have = 0;
                }
            }
            seen += 1;
        }
    }
    if (have > 0) flush(have);
#pragma unroll
    for (int m = 0; m < NCH; ++m) {
        const int r = lane * 2 + 128 * m;
        if (r < N) *reinterpret_cast<double2 *>(stage + (size_t)which * N + r) = acc[m];
    }
    return seen;
}


// zB[j] changed by dz (a variable entered B at a nonzero value, or left it): the cached hq = V[:,nz(zB)] zB + q and
// bEall = rhs - [A;G] zB follow by ONE column each (hq += V[:,j] dz, bEall -= [A;G][:,j] dz) instead of being
// re-evaluated from every nonzero column / every constraint row in the next pass.  All threads; N even, N <= 512.
__device__ __forceinline__ void bound_event(const Lds &L, const double *__restrict__ V, const double *__restrict__ Ct,
                                            int N, int MJ, int j, double dz) {
    const int tid = threadIdx.x;
    const int r = 2 * tid;
    const double2 v = *reinterpret_cast<const double2 *>(V + (size_t)j * N + (r < N ? r : 0));
    const double cj = Ct[(size_t)(tid < MJ ? tid : 0) * N + j];
    if (r < N) {
        double2 hv = *reinterpret_cast<double2 *>(L.hq + r);
        hv.x = fma(v.x, dz, hv.x);
        hv.y = fma(v.y, dz, hv.y);
        *reinterpret_cast<double2 *>(L.hq + r) = hv;
    }
    if (tid < MJ) L.bEall[tid] = fma(-cj, dz, L.bEall[tid]);
    if (tid == 0) {
        L.ired[HQ_CHANGED] = 1;  // the border column L^-1 c kept from the last pass is stale
        // rounding of these one-column updates must not pile up over a long run (the reference re-evaluates
        // VBF'zB and bE in every pass): every 64th switch asks for a full re-evaluation
        if (++L.ired[SHIFT_COUNT] >= 64) {
            L.ired[SHIFT_COUNT] = 0;
            L.ired[HB_DIRTY] = 1;
        }
    }
}

// ------------------------------------------------------------------ the loop
struct ProbCtx {
    int N, M, J, MJ;
    double tol, tolG;
    const double *__restrict__ V;
    const double *__restrict__ Ct;   // row r of [A;G] contiguous
    const double *__restrict__ rhs;  // [b; g]
    const double *__restrict__ q;
    const double *__restrict__ dlo;
    const double *__restrict__ uhi;
    ssqp_trace *trace;
    int ntrace;
    bool wantMult;    // multiplier outputs were asked for: the last pass leaves what solve_one needs in LDS
    int arenaCap;
    bool dense;       // read zero-weight columns too (roofline measurement of the dense formulation)
    double *garena;
    // results / accounting
    int64_t iter, ret;
    int32_t det;
    int64_t sBytes, sRead, sFlops, sK3;
    int maxK, pathBits;
    // incremental engine state
    int Kfac;     // rows currently in the kept factor (0: empty / invalid)
    int RC;       // factor capacity (0: engine disabled for this problem shape)
    int yOff, rdvOff, facOff;  // offsets (doubles) of Y, 1/d and the factor in their arena
    int scrCap;                // doubles of scratch in front of them (LDS arena)
    double qr0, qr1;           // this thread's two entries of q, requested early for the refresh of hq (per thread)
    int fwdR0;                 // first row this pass still has to forward-substitute (front half)
    bool fwdRows;              // ... row by row (few new rows) instead of a sweep over the factor
    int yValid, yW0;           // leading rows of the border columns Y that are already forward-substituted for the
                               // current factor, and the row count W0 they were formed for (front half)
    bool hbValid;              // L.hq holds hB + q of the current bound set (front half)
    bool facGlobal;            // engine state (factor, 1/d, Y) lives in the workgroup's GLOBAL arena (K grew past the LDS capacity)
#ifdef SSQP_PHASE_PROFILE
    unsigned long long ph_last;
    int ph_cur;
#endif
};

enum { ACT_CONTINUE = 0, ACT_BREAK = 1 };

// One pass of the loop for K > 0, from the E-row sweep to the status switch.
// INLDS selects where the arena (X, packed factor, Schur block) lives: the
// instantiation with INLDS=true only ever sees LDS pointers.
// FG: the kept factor, its reciprocal pivots and the border right-hand sides live in the global arena (large K);
// the scratch (X, AXPY staging, Schur block) stays in LDS.
template <int VEC, bool INLDS, bool FG>
__device__ __forceinline__ int iterate_kkt(ProbCtx &C, const Lds &L, double *ar, int K, int W0, int JO) {
    const int N = C.N, M = C.M, J = C.J, MJ = C.MJ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double tol = C.tol, tolG = C.tolG;
    const double *__restrict__ V = C.V;
    const double *__restrict__ Ct = C.Ct;
    const double *__restrict__ rhs = C.rhs;
    const double *__restrict__ q = C.q;
    const double *__restrict__ dlo = C.dlo;
    const double *__restrict__ uhi = C.uhi;
    const double inf = __longlong_as_double(0x7ff0000000000000ll);
    const int R = N - K, JE = W0 - M;
    // per-thread slots over the free list: N <= 512 needs 2 at 256 threads (VEC 2), N <= 1024 4, else 8
    constexpr int MPT = (VEC == 2) ? 2 : ((VEC == 3) ? 4 : MAXPT);
    const int64_t iter = C.iter;
    ssqp_trace *trace = (C.trace && iter <= C.ntrace) ? C.trace + (iter - 1) : nullptr;

    // bounds of this thread's free variables for the ratio test: requested now, needed only in aStep! -- the
    // gather's memory latency is hidden behind the whole KKT solve
    double ureg[MPT], dreg[MPT];
#pragma unroll
    for (int m = 0; m < MPT; ++m) {
        const int k = tid + m * NT;
        const int i = L.idx[k < K ? k : 0];
        ureg[m] = uhi[i];
        dreg[m] = dlo[i];
    }
    // ---- incremental engine: bring the kept factor in line with the current free set ----
    bool useInc = false;
    if (INLDS) useInc = (C.RC > 0) && (K <= C.RC) && (W0 + 1 <= INC_RBM);
    Inc I;
    {
        double *eb = FG ? C.garena : ar;  // base of the engine state
        I.fcol = eb + C.facOff;
        I.rdv = eb + C.rdvOff;
        I.Y = eb + C.yOff;
    }
    I.ord = L.ordl;
    I.fpos = L.fpos;
    I.RC = C.RC;
    // wavefront-specialised front half (see inc_sync_w0): needs the AXPY loads (N even, N <= 512), a rank filter
    // that fits in registers, and room in the scratch arena for two partial vectors plus X
    bool frontDone = false;
    bool cachedRows = false;  // (byte accounting: the constraint rows are not swept in this pass)
    int Wspec = W0;
    if (useInc && VEC == 2 && W0 <= RF_ROWS && (W0 <= 4 ? K + 1 <= 256 : K + 1 <= 192) &&
        (long)2 * N + (long)W0 * (K + 1) + 8 <= C.scrCap) {
        PHASE(C, 1);
        const bool needHB = C.dense || !C.hbValid || (L.ired[HB_DIRTY] != 0);
        const bool hqChanged = L.ired[HQ_CHANGED] != 0;  // (read here, reset after the join: barriers in between)
        // do the kept border columns Y still belong to the same constraint rows, slot by slot?
        bool tagsSame = (W0 == C.yW0);
        {
            const bool bad = (lane < W0) && (L.rowsE[lane < W0 ? lane : 0] != L.ytag[lane < 16 ? lane : 0]);
            if (__ballot(bad) != 0ull) tagsSame = false;
        }
        double *X = ar + 2 * N;
        // E rows (SSQP.jl:290-295).  bE = b - [A;G] zB only changes with zB, like hB: when every row fits the batch
        // (MJ <= 12) it is kept for ALL rows next to hq and re-evaluated only on needHB; the entries X needs,
        // AE = [A;G][E, F], are then a K x W0 gather (one element per thread) instead of full-row sweeps.
        const bool cacheBE = (MJ <= RF_ROWS);
        if (cacheBE) {
            // AE entries for X: one 8-byte gather per thread and trip (16 slots x 16 columns, no integer division);
            // the first trip's load is issued before the row sweep below, so both wait on memory together
            const int wX = tid & 15;
            const int ridX = L.rowsE[wX < W0 ? wX : 0];
            const int kX = tid >> 4;
            double x0v = 0.0;
            if (wX < W0 && kX < K) x0v = Ct[(size_t)ridX * N + L.idx[kX]];
            // q for the refresh of hq after the join (needed ~10 k cycles from now)
            double qreg[MPT];
#pragma unroll
            for (int m = 0; m < MPT; ++m) {
                const int i = tid + m * NT;
                qreg[m] = (needHB && i < N) ? q[i] : 0.0;
            }
            if (needHB && MJ > 0) {
                constexpr int RPW = (RF_ROWS + NW - 1) / NW;
                const double *__restrict__ rows[RPW];
                double rh[RPW];
#pragma unroll
                for (int t = 0; t < RPW; ++t) {
                    const int rid = wave + NW * t;
                    const int rc = rid < MJ ? rid : 0;
                    rows[t] = (rid < MJ || t == 0) ? Ct + (size_t)rc * N : nullptr;
                    rh[t] = rhs[rc];
                }
                double a1[RPW], a2[RPW];
                rows_dot2_batch<RPW>(rows, L.zm, L.zm, N, lane, a1, a2);
#pragma unroll
                for (int t = 0; t < RPW; ++t) {
                    const int rid = wave + NW * t;
                    if (rid < MJ) {  // uniform per wavefront
                        const double acc = wave_sum(a1[t]);
                        if (lane == 0) L.bEall[rid] = rh[t] - acc;
                    }
                }
            }
            if (wX < W0 && kX < K) X[wX + W0 * kX] = x0v;
            for (int k = kX + NT / 16; k < K; k += NT / 16)
                if (wX < W0) X[wX + W0 * k] = Ct[(size_t)ridX * N + L.idx[k]];
            C.qr0 = qreg[0];
            C.qr1 = (MPT > 1) ? qreg[MPT > 1 ? 1 : 0] : 0.0;
            if (needHB) __syncthreads();  // (bEall of this pass)
            if (tid < W0) {
                const double be = L.bEall[L.rowsE[tid]];
                L.bE[tid] = be;
                X[tid + W0 * K] = be;
            }
        } else
        if (W0 > 0) {  // E-row sweep (SSQP.jl:290-295), rows over the wavefronts, every load of a wavefront's rows in flight
            // before the first use (W0 <= 12: at most three rows each)
            constexpr int RPW = (RF_ROWS + NW - 1) / NW;
            double2 g[RPW][4];
            double rh[RPW];
#pragma unroll
            for (int t = 0; t < RPW; ++t) {
                const int w = wave + NW * t;
                const int r = L.rowsE[w < W0 ? w : 0];
                const double *__restrict__ row = Ct + (size_t)r * N;
                rh[t] = rhs[r];
                if (w < W0 || t == 0) {  // uniform
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int i = lane * 2 + 128 * m;
                        g[t][m] = *reinterpret_cast<const double2 *>(row + (i < N ? i : 0));
                    }
                }
            }
            double2 zz[4];
            int p0[4], p1[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int i = lane * 2 + 128 * m;
                const int ic = i < N ? i : 0;
                zz[m] = *reinterpret_cast<const double2 *>(L.zm + ic);
                p0[m] = (i < N) ? (int)L.pos[ic] : -1;
                p1[m] = (i < N) ? (int)L.pos[ic + 1] : -1;
            }
#pragma unroll
            for (int t = 0; t < RPW; ++t) {
                const int w = wave + NW * t;
                if (w < W0) {  // uniform
                    double acc = 0.0;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int i = lane * 2 + 128 * m;
                        const double na = fma(g[t][m].y, zz[m].y, fma(g[t][m].x, zz[m].x, acc));
                        acc = (i < N) ? na : acc;
                        if (p0[m] >= 0) X[w + W0 * p0[m]] = g[t][m].x;
                        if (p1[m] >= 0) X[w + W0 * p1[m]] = g[t][m].y;
                    }
                    acc = wave_sum(acc);
                    if (lane == 0) {
                        const double be = rh[t] - acc;
                        L.bE[w] = be;
                        X[w + W0 * K] = be;
                    }
                }
            }
        }
        __syncthreads();
        PHASE(C, 13);
        if (wave == 0) {
            SUBPHASE_DECL(tw0);
            int nApp = 0, minDel = 0;
            const int Kf = inc_sync_w0(I, C.Kfac, L, V, N, K, nApp, minDel);
            if (lane == 0) {
                L.ired[2 * NW + 10] = Kf;
                L.ired[2 * NW + 11] = nApp;
                L.ired[2 * NW + 14] = minDel;
            }
            SUBPHASE(14, tw0);
        } else if (wave == 1) {
            SUBPHASE_DECL(tw1);
            int w = W0;
            if (W0 > 0) {
                if (W0 <= 4 && K + 1 <= 128) w = rank_filter_regs<4, 2>(X, W0, K + 1, tol, L);
                else if (W0 <= 4) w = rank_filter_regs<4, 4>(X, W0, K + 1, tol, L);
                else if (K + 1 <= 64 && W0 <= 8) w = rank_filter_regs<8, 1>(X, W0, K + 1, tol, L);
                else if (K + 1 <= 64) w = rank_filter_regs<RF_ROWS, 1>(X, W0, K + 1, tol, L);
                else if (K + 1 <= 128) w = rank_filter_regs<RF_ROWS, 2>(X, W0, K + 1, tol, L);
                else w = rank_filter_wave(X, W0, K + 1, tol, L);
            }
            if (w < W0) {
                const double v = (lane < w) ? L.bE[L.ra[lane < w ? lane : 0]] : 0.0;
                wave_sync();
                if (lane < w) L.bE[lane] = v;
            } else if (lane < W0) {
                L.ra[lane] = (int16_t)lane;
            }
            if (lane == 0) L.ired[2 * NW + 4] = w;
            WSTAMP(28, tw1);
        } else {
            SUBPHASE_DECL(tw2);
            // hB = V[:, nz(zB)] zB only changes when a bound variable with z != 0 enters or leaves B: otherwise the
            // cached hq = hB + q of the last evaluation is reused (same bits: same inputs, same operations)
            int cnt = 0;
            if (needHB) cnt = hb_partial<4>(V, N, L.zm, wave - 2, ar);
            if (wave == 2 && lane == 0) L.ired[2 * NW + 12] = cnt;
            if (wave == 2) WSTAMP(29, tw2);
        }
        __syncthreads();
        PHASE(C, 3);
        const int Kf = uni(L.ired[2 * NW + 10]);
        C.Kfac = Kf;
        if (Kf < 0) {  // cholesky(V[F,F]) of the reference would throw here
            C.ret = -1;
            C.det = SSQP_DETAIL_POSDEF_V;
            return ACT_BREAK;
        }
        Wspec = L.ired[2 * NW + 4];
        C.sRead += 64ll * L.ired[2 * NW + 11] * K + 8ll * N * L.ired[2 * NW + 12] - 8ll * N * K;
        if (MJ <= RF_ROWS) {  // E rows: a W0 x K gather (one 64-byte sector per entry); the full rows (and q) are swept
                              // only when bE / hq are re-evaluated.  (The common line below adds 8*MJ*N: taken back
                              // here; aStep! adds the rows it reads.)
            C.sRead += 64ll * W0 * K - 8ll * MJ * N + (needHB ? 8ll * MJ * N + 8ll * N : 0ll);
            cachedRows = true;
        }
        // border right-hand sides AE' in factor order: from X in LDS when the register rank filter left it intact,
        // else (the LDS filter eliminated in place) gathered again from the constraint rows
        const bool xKept = !(W0 > 4 && K + 1 > 128);
        // Forward substitution is causal (row r of L^-1 b needs rows <= r only): the entries of Y computed in the
        // last pass stay valid for every row below the lowest deleted one, as long as the right-hand sides are
        // the same (same constraint rows in the same slots, same hq).  After an append only the new row is missing.
        int r0 = C.yValid;
        {
            const int minDel = uni(L.ired[2 * NW + 14]);
            if (minDel < r0) r0 = minDel;
            if (needHB || !tagsSame || r0 > K || hqChanged) r0 = 0;
        }
        const bool rowMode = (r0 > 0) && (K - r0 <= 2);
        if (!rowMode) r0 = 0;
        C.fwdR0 = r0;
        C.fwdRows = rowMode;
        {
            const int nr = K - r0;  // raw right-hand sides of the rows still to be substituted
            for (int e = tid; e < W0 * nr; e += NT) {
                const int w = e / nr, r = r0 + (e - w * nr);
                const int iv = I.ord[r];
                I.Y[(size_t)w * I.RC + r] = xKept ? X[w + W0 * (int)L.pos[iv]] : Ct[(size_t)L.rowsE[w] * N + iv];
            }
            for (int r = r0 + tid; r < K; r += NT) {  // c = hB[F] + q[F]; hB = sum of the two partial vectors
                const int i = I.ord[r];
                I.Y[(size_t)W0 * I.RC + r] = needHB ? (ar[i] + ar[N + i]) + q[i] : L.hq[i];
            }
        }
        if (tid < W0 && tid < 16) L.ytag[tid] = L.rowsE[tid];
        C.yW0 = W0;
        C.yValid = K;
        if (needHB) {
            if (MJ <= RF_ROWS) {  // (q was requested before the E pass: see cacheBE)
                if (tid < N) L.hq[tid] = (ar[tid] + ar[N + tid]) + C.qr0;
                if (tid + NT < N) L.hq[tid + NT] = (ar[tid + NT] + ar[N + tid + NT]) + C.qr1;
            } else {
                for (int i = tid; i < N; i += NT) L.hq[i] = (ar[i] + ar[N + i]) + q[i];
            }
        }
        if (tid == 0) {
            L.ired[HB_DIRTY] = 0;
            L.ired[HQ_CHANGED] = 0;
        }
        C.hbValid = true;
        if (tid <= Wspec) L.perm[tid] = (tid < Wspec) ? L.ra[tid] : (int16_t)W0;  // physical right-hand sides
        __syncthreads();
        frontDone = true;
    }
    if (useInc && !frontDone) {
        PHASE(C, 13);
        if (wave == 0) {
            int nApp = 0, minDel = 0;
            const int Kf = inc_sync_w0(I, C.Kfac, L, V, N, K, nApp, minDel);
            if (lane == 0) {
                L.ired[2 * NW + 10] = Kf;
                L.ired[2 * NW + 11] = nApp;
            }
        }
        C.yValid = 0;  // (this path refills and re-substitutes the border columns from scratch)
        __syncthreads();
        C.Kfac = uni(L.ired[2 * NW + 10]);
        C.sRead += 64ll * L.ired[2 * NW + 11] * K;  // the K scattered entries of each appended column
        if (C.Kfac < 0) {  // cholesky(V[F,F]) of the reference would throw here
            C.ret = -1;
            C.det = SSQP_DETAIL_POSDEF_V;
            return ACT_BREAK;
        }
    }

    // ---- E-row sweep: bE and X = [AE bE]   (SSQP.jl:290-295) ----
    if (!frontDone) PHASE(C, 1);
    if (!frontDone) {
        double *X = ar;
        for (int w = wave; w < W0; w += NW) {
            const int r = L.rowsE[w];
            const double *__restrict__ row = Ct + (size_t)r * N;
            double acc = 0.0;
            if (VEC >= 3) {  // (large N: eight 16-byte loads per lane in flight; the sum in the same order)
                for (int i0 = lane * 2; i0 < N; i0 += 128 * 8) {
                    double2 v8[8];
#pragma unroll
                    for (int m = 0; m < 8; ++m) v8[m] = *reinterpret_cast<const double2 *>(row + (i0 + 128 * m < N ? i0 + 128 * m : lane * 2));
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const int i = i0 + 128 * m;
                        if (i < N) {
                            const double2 zz = *reinterpret_cast<const double2 *>(L.zm + i);
                            const int p0 = L.pos[i], p1 = L.pos[i + 1];
                            acc = fma(v8[m].y, zz.y, fma(v8[m].x, zz.x, acc));
                            if (p0 >= 0) X[w + W0 * p0] = v8[m].x;
                            if (p1 >= 0) X[w + W0 * p1] = v8[m].y;
                        }
                    }
                }
            } else if (VEC >= 2) {
#pragma unroll 4
                for (int i = lane * 2; i < N; i += 128) {
                    const double2 v = *reinterpret_cast<const double2 *>(row + i);
                    const double2 zz = *reinterpret_cast<const double2 *>(L.zm + i);
                    const int p0 = L.pos[i], p1 = L.pos[i + 1];
                    acc = fma(v.y, zz.y, fma(v.x, zz.x, acc));
                    if (p0 >= 0) X[w + W0 * p0] = v.x;
                    if (p1 >= 0) X[w + W0 * p1] = v.y;
                }
            } else {
#pragma unroll 4
                for (int i = lane; i < N; i += 64) {
                    const double v = row[i];
                    acc = fma(v, L.zm[i], acc);
                    const int p = L.pos[i];
                    if (p >= 0) X[w + W0 * p] = v;
                }
            }
            acc = wave_sum(acc);
            if (lane == 0) {
                const double be = rhs[r] - acc;
                L.bE[w] = be;
                X[w + W0 * K] = be;
            }
        }
    }
    if (!frontDone) __syncthreads();
    if (useInc && !frontDone) {  // border right-hand sides AE' in factor order (the rank filter is about to destroy X)
        for (int e = tid; e < W0 * K; e += NT) {
            const int w = e / K, r = e - w * K;
            I.Y[(size_t)w * I.RC + r] = ar[w + W0 * (int)L.pos[I.ord[r]]];
        }
        __syncthreads();
    }
    // ---- rank filter  (SSQP.jl:310-319) ----
    if (!frontDone) PHASE(C, 2);
    int W = frontDone ? Wspec : W0;
    if (W0 > 0 && !frontDone) {
        if ((long)W0 * (K + 1) <= 4096 && K + 1 <= 64 * RF_CS) {  // small: one wavefront, no workgroup barriers
            if (wave == 0) {
                SUBPHASE_DECL(trf);
                int w;
                if (W0 <= 4 && K + 1 <= 128) w = rank_filter_regs<4, 2>(ar, W0, K + 1, tol, L);
                else if (W0 <= RF_ROWS && K + 1 <= 64 * RF_CS) w = rank_filter_regs<RF_ROWS, RF_CS>(ar, W0, K + 1, tol, L);
                else w = rank_filter_wave(ar, W0, K + 1, tol, L);
                if (lane == 0) L.ired[2 * NW + 4] = w;
                SUBPHASE(14, trf);
            }
            __syncthreads();
            W = L.ired[2 * NW + 4];
        } else {
            W = rank_filter(ar, W0, K + 1, tol, L);
        }
    }
    if (!frontDone) {
        if (W < W0) {
            double v = 0.0;
            if (tid < W) v = L.bE[L.ra[tid]];
            __syncthreads();
            if (tid < W) L.bE[tid] = v;
        } else {
            for (int w = tid; w < W0; w += NT) L.ra[w] = (int16_t)w;
        }
        __syncthreads();
    }

    // ratio-test rows fetched ahead by the wavefronts that idle during the lambda / alpha solve (see below)
    constexpr int PRE = 2;       // rows per prefetching wavefront (wavefronts 1..3: up to 6 inactive rows)
    double2 gpre[PRE][4];
    double azpre[PRE], rhpre[PRE];
    int npre = 0;
#pragma unroll
    for (int t = 0; t < PRE; ++t) {
        azpre[t] = 0.0;
        rhpre[t] = 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m) gpre[t][m] = make_double2(0.0, 0.0);
    }
    double *fac = ar;  // (from-scratch path: packed factor; least-squares scratch of KKTchk! in both paths)
    if (useInc) {
        // =========================== incremental engine ===========================
        PHASE(C, 3);
        C.pathBits |= 4;
        // c = V[B,F]'zB + q[F]: hB = V[:, nz(zB)] zB in AXPY form (only the bound variables that are not at 0)
        if (!frontDone) {
            const int ncols = stream_matvec<VEC>(V, N, L.zm, C.dense, ar, L.gam, L);
            C.sRead += 8ll * N * ncols - 8ll * N * K;  // (the K term is added by the common accounting below)
            __syncthreads();
        }
        if (!frontDone) {
            for (int r = tid; r < K; r += NT) {
                const int i = I.ord[r];
                I.Y[(size_t)W0 * I.RC + r] = L.gam[i] + q[i];
            }
            if (tid <= W) L.perm[tid] = (tid < W) ? L.ra[tid] : (int16_t)W0;  // physical right-hand sides
            __syncthreads();
        }
        PHASE(C, 4);
        if (frontDone && C.fwdRows) {  // only the rows appended since the last pass
            if (C.fwdR0 < K)
                INC_BY_SLOTS(K, inc_forward_rows<1>(I, K, C.fwdR0, W0 + 1), inc_forward_rows<2>(I, K, C.fwdR0, W0 + 1),
                             inc_forward_rows<4>(I, K, C.fwdR0, W0 + 1));
        } else {
            // front half: every slot (all W0 rows and c), so that the columns stay valid for the next pass
            const int16_t *phys = frontDone ? nullptr : L.perm;
            const int nrhs = frontDone ? W0 + 1 : W + 1;
            if (K <= 64) {
                if (nrhs > 2 * NW) inc_forward_border<1, 3>(I, K, nrhs, phys);
                else inc_forward_border<1, 2>(I, K, nrhs, phys);
            } else if (K <= 128) {
                inc_forward_border<2, 2>(I, K, nrhs, phys);
            } else {
                inc_forward_border<4, 2>(I, K, nrhs, phys);
            }
        }
        __syncthreads();
        PHASE(C, 5);
        // Schur block: H = AE V^-1 AE' (W x W, lower) into the scratch arena, t = AE V^-1 c into tv.
        // Four lanes per (a, b) pair, each walking every fourth of the K rows, then a two-step DPP sum inside
        // the quad; all pairs in parallel (W <= 11: at most 77 pairs over the 64 quads).
        double *H = ar;
        {
            const int npairs = W * (W + 1) / 2 + W;
            const int qd = tid >> 2, ql = tid & 3;
            for (int e0 = 0; e0 < npairs; e0 += NT / 4) {
                const int e = (e0 + qd < npairs) ? e0 + qd : 0;
                int a, b;
                if (e < W) {
                    a = W;
                    b = e;
                } else {
                    int f = e - W, col = 0;
                    while (f >= W - col) {
                        f -= W - col;
                        ++col;
                    }
                    a = col + f;
                    b = col;
                }
                const double *Ya = I.Y + (size_t)L.perm[a] * I.RC, *Yb = I.Y + (size_t)L.perm[b] * I.RC;
                double acc = 0.0;
#pragma unroll 4
                for (int r = ql; r < K; r += 4) acc = fma(Ya[r] * I.rdv[r], Yb[r], acc);
                acc += dpp_f64<DPP_XOR1>(acc);
                acc += dpp_f64<DPP_XOR2>(acc);
                if (ql == 0 && e0 + qd < npairs) {
                    if (e < W) L.tv[b] = acc;
                    else H[a + W * b] = acc;
                }
            }
        }
        __syncthreads();
        SUBPHASE_DECL(tlam);
        if (wave == 0) {  // lambda: H lam = bE + t, alphaL = -lam (W <= 11, in registers); then alpha, no barrier between
            if (lane < W) L.tv[lane] = L.bE[lane] + L.tv[lane];
            wave_sync();
            double lam = 0.0;
            bool okH = true;
            double *tr = H + W * W;
            if (W > 8) okH = small_spd_solve<INC_RBM - 1>(H, L.tv, W, lam, tr);
            else if (W > 4) okH = small_spd_solve<8>(H, L.tv, W, lam, tr);
            else if (W > 0) okH = small_spd_solve<4>(H, L.tv, W, lam, tr);
            if (lane < W) L.aL[lane] = -lam;
            if (lane == 0) L.ired[2 * NW + 2] = okH ? 1 : 0;
            wave_sync();
            SUBPHASE(19, tlam);
            if (okH)  // v = D^-1 (Y_A alphaL + y_c); alpha = -L'^-1 v
                INC_BY_SLOTS(K, inc_alpha<1>(I, K, W, W0, L.perm, L.aL, L.gam),
                             inc_alpha<2>(I, K, W, W0, L.perm, L.aL, L.gam),
                             inc_alpha<4>(I, K, W, W0, L.perm, L.aL, L.gam));
        } else if (VEC == 2) {
            // meanwhile the other wavefronts fetch the rows of the inactive inequalities for aStep! (G z is formed
            // now, G p once p exists): the memory round trip of the ratio test is off the critical path
            npre = (JO < 3 * PRE) ? JO : 3 * PRE;
#pragma unroll
            for (int t = 0; t < PRE; ++t) {
                const int o = (wave - 1) + 3 * t;
                if (o < npre) {  // uniform per wavefront
                    const int j = L.iO[o];
                    const double *__restrict__ row = Ct + (size_t)(M + j) * N;
                    rhpre[t] = rhs[M + j];
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int r = lane * 2 + 128 * m;
                        gpre[t][m] = *reinterpret_cast<const double2 *>(row + (r < N ? r : 0));
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < PRE; ++t) {
                const int o = (wave - 1) + 3 * t;
                if (o < npre) {
                    double s1 = 0.0;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int r = lane * 2 + 128 * m;
                        const double2 zz = *reinterpret_cast<const double2 *>(L.z + (r < N ? r : 0));
                        const double n1 = fma(gpre[t][m].y, zz.y, fma(gpre[t][m].x, zz.x, s1));
                        s1 = (r < N) ? n1 : s1;
                    }
                    azpre[t] = s1;
                }
            }
        }
        if (VEC == 2) npre = (JO < 3 * PRE) ? JO : 3 * PRE;
        __syncthreads();
        if (!L.ired[2 * NW + 2]) {
            C.ret = -1;
            C.det = SSQP_DETAIL_POSDEF_C;
            return ACT_BREAK;
        }
        PHASE(C, 6);
    } else {
        C.Kfac = -1;  // the from-scratch path overwrites the arena: the kept factor is gone
        C.yValid = 0;
    // ---- factor assembly: pass 1 over V[:,F] + border rows ----
    PHASE(C, 3);
    const int Rr = K + W + 1;
    double *rd = ar + coloff(Rr, Rr);  // Rr reciprocal pivots behind the packed R x R triangle
    for (int t = wave * 2; t < K; t += NW * 2) {
        const bool two = (t + 1 < K);
        const int k0 = t, k1 = two ? t + 1 : t;
        const int j0 = L.idx[k0], j1 = L.idx[k1];
        const int o0 = coloff(k0, Rr), o1 = coloff(k1, Rr);
        double a0, a1;
        dot2_cols_gather<VEC>(V + (size_t)j0 * N, V + (size_t)j1 * N, L.zm, L.pos, fac, o0, o1, k0, k1, two, N, lane,
                              a0, a1);
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        if (lane == 0) {  // c = V[B,F]'zB + q[F]   (SSQP.jl:324)
            fac[o0 + (K + W) - k0] = a0 + q[j0];
            if (two) fac[o1 + (K + W) - k1] = a1 + q[j1];
        }
    }
    for (int e = tid; e < W * K; e += NT) {  // AE rows (kept ones)
        const int w = e / K, k = e - w * K;
        const int r = L.rowsE[L.ra[w]];
        fac[coloff(k, Rr) + (K + w) - k] = Ct[(size_t)r * N + L.idx[k]];
    }
    {  // zero border-border block (columns K..K+W are contiguous at the end of the packed triangle)
        const int ob = coloff(K, Rr), ne = coloff(Rr, Rr) - ob;
        for (int e = tid; e < ne; e += NT) fac[ob + e] = 0.0;
    }
    // (barrier at the top of bordered_ldl)
#ifdef SSQP_PHASE_PROFILE
    __syncthreads();
#endif
    PHASE(C, 4);
    if (!bordered_ldl(fac, rd, 0, K, Rr, L)) {
        C.ret = -1;
        C.det = SSQP_DETAIL_POSDEF_V;
        return ACT_BREAK;
    }
    // ---- Schur system (AE V^-1 AE') lam = bE + AE V^-1 c ; alphaL = -lam  (SSQP.jl:325-328, 351) ----
    PHASE(C, 5);
    if (W > 0) {
        // trailing block holds -H (W x W) and, in the c row, -t: flip the sign of H, put s = bE + t in the row
        for (int e = tid; e < W * (W + 1) / 2 + W; e += NT) {
            if (e < W) {
                const int o = coloff(K + e, Rr) + (W - e);
                fac[o] = L.bE[e] - fac[o];
            } else {
                int f = e - W, col = 0;
                while (f >= W - col) {
                    f -= W - col;
                    ++col;
                }
                const int o = coloff(K + col, Rr) + f;
                fac[o] = -fac[o];
            }
        }
        if (!bordered_ldl(fac, rd, K, K + W, Rr, L)) {
            C.ret = -1;
            C.det = SSQP_DETAIL_POSDEF_C;
            return ACT_BREAK;
        }
        if (wave == 0) {
            for (int w = lane; w < W; w += 64) L.aL[w] = fac[coloff(K + w, Rr) + (W - w)] * rd[K + w];
            wave_sync();
            back_substitute(fac, rd, L.aL, K, W, Rr);
            for (int w = lane; w < W; w += 64) L.aL[w] = -L.aL[w];
        }
        __syncthreads();
    }
    // v = D^-1 (Y_A alphaL + y_c); alpha = -L'^-1 v
    PHASE(C, 6);
    double *vk = L.gam;
    for (int j = tid; j < K; j += NT) {
        const int oj = coloff(j, Rr);
        double s = fac[oj + K + W - j];
        for (int w = 0; w < W; ++w) s = fma(fac[oj + K + w - j], L.aL[w], s);
        vk[j] = s * rd[j];
    }
    __syncthreads();
    if (wave == 0) back_substitute(fac, rd, vk, 0, K, Rr);
    __syncthreads();
    {  // alpha = -x, scattered to the variables' own slots of gam
        double t[MPT];
#pragma unroll
        for (int m = 0; m < MPT; ++m) {
            const int k = tid + m * NT;
            t[m] = (k < K) ? -vk[k] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MPT; ++m) {
            const int k = tid + m * NT;
            if (k < K) L.gam[L.idx[k]] = t[m];
        }
        __syncthreads();
    }
    }
    // ---- alpha (scattered into gam), p (scattered into zm) ----
    PHASE(C, 7);
    double areg[MPT];
#pragma unroll
    for (int m = 0; m < MPT; ++m) {
        const int k = tid + m * NT;
        areg[m] = (k < K) ? L.gam[L.idx[k < K ? k : 0]] : 0.0;
    }
    double pa = 0.0;
    int pnan = 0;
    for (int i = tid; i < N; i += NT)
        if (L.pos[i] < 0) L.zm[i] = 0.0;
#pragma unroll
    for (int m = 0; m < MPT; ++m) {
        const int k = tid + m * NT;
        if (k < K) {
            const int i = L.idx[k];
            const double p = areg[m] - L.z[i];
            L.zm[i] = p;
            if (p != p) pnan = 1;
            pa = fmax(pa, fabs(p));
        }
    }
    double pinf;
    int anyNan;
    {   // max |p| and "any NaN" in one exchange (its barriers also order the writes above)
        const double wm = wave_max(pa);
        const unsigned long long nb = __ballot(pnan);
        if (lane == 0) {
            L.red[wave] = wm;
            L.ired[wave] = (nb != 0ull);
        }
        __syncthreads();
        double r = L.red[0];
        int f = L.ired[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            r = fmax(r, L.red[w]);
            f |= L.ired[w];
        }
        __syncthreads();
        pinf = r;
        anyNan = f;
    }

    // per-pass accounting (the R^2 terms only when gamma is formed)
    {
        const long long k = K, r = R, w = W;
        C.sBytes += 8ll * (k * k + r * k) + 8ll * MJ * N + 48ll * N + 4ll * (N + J);
        C.sRead += 8ll * N * k + 8ll * MJ * N + 48ll * N + 4ll * (N + J);
        C.sFlops += k * k * k + 4 * k * k * w + 2 * k * k + 2 * r * k + 2ll * W0 * r + w * w * w;
        C.sK3 += k * k * k;
    }

    // the caches hq / bEall are live (front half, all rows cached): status switches update them by one column
    const bool evtOK = (VEC == 2) && frontDone && !C.dense && C.hbValid && (MJ <= RF_ROWS);
    PHASE(C, 8);
    if (cachedRows && pinf > tolG && !anyNan) C.sRead += 8ll * N * JO;  // rows of the inactive inequalities
    if (pinf > tolG && !anyNan) {  // ------------------------ aStep!  SSQP.jl:61-134
        // inactive inequalities: zo = g - G z, po = G[:,F] p   (:78-89)
        SUBPHASE_DECL(tast);
        if (VEC == 2) {
            if (wave > 0) {  // rows fetched during the lambda / alpha solve
#pragma unroll
                for (int t = 0; t < PRE; ++t) {
                    const int o = (wave - 1) + 3 * t;
                    if (o < npre) {  // uniform per wavefront
                        double s2 = 0.0;
#pragma unroll
                        for (int m = 0; m < 4; ++m) {
                            const int r = lane * 2 + 128 * m;
                            const double2 pp = *reinterpret_cast<const double2 *>(L.zm + (r < N ? r : 0));
                            const double n2 = fma(gpre[t][m].y, pp.y, fma(gpre[t][m].x, pp.x, s2));
                            s2 = (r < N) ? n2 : s2;
                        }
                        const double sz = wave_sum(azpre[t]), sp = wave_sum(s2);
                        if (lane == 0) L.lin[o] = (sp > tol) ? (rhpre[t] - sz) / sp : inf;
                    }
                }
            }
            constexpr int RPW = 3;  // rows per wavefront and batch
            for (int o0 = npre; o0 < JO; o0 += NW * RPW) {
                const double *__restrict__ rows[RPW];
                double rh[RPW];
#pragma unroll
                for (int t = 0; t < RPW; ++t) {
                    const int o = o0 + wave + NW * t;
                    const int j = L.iO[o < JO ? o : o0];
                    rows[t] = (o < JO || t == 0) ? Ct + (size_t)(M + j) * N : nullptr;
                    rh[t] = rhs[M + j];
                }
                double az[RPW], ap[RPW];
                rows_dot2_batch<RPW>(rows, L.z, L.zm, N, lane, az, ap);
#pragma unroll
                for (int t = 0; t < RPW; ++t) {
                    const int o = o0 + wave + NW * t;
                    if (o < JO) {  // uniform per wavefront
                        const double sz = wave_sum(az[t]), sp = wave_sum(ap[t]);
                        if (lane == 0) L.lin[o] = (sp > tol) ? (rh[t] - sz) / sp : inf;
                    }
                }
            }
        } else {
            if (VEC >= 3) {  // (large N: two rows per wavefront and round)
                for (int o = wave; o < JO; o += 2 * NW) {
                    const int o2 = o + NW < JO ? o + NW : o;
                    const int j = L.iO[o], j2 = L.iO[o2];
                    double az, ap, az2, ap2;
                    rows2_dot2(Ct + (size_t)(M + j) * N, Ct + (size_t)(M + j2) * N, L.z, L.zm, N, lane, az, ap, az2, ap2);
                    az = wave_sum(az);
                    ap = wave_sum(ap);
                    az2 = wave_sum(az2);
                    ap2 = wave_sum(ap2);
                    if (lane == 0) {
                        L.lin[o] = (ap > tol) ? (rhs[M + j] - az) / ap : inf;
                        if (o + NW < JO) L.lin[o + NW] = (ap2 > tol) ? (rhs[M + j2] - az2) / ap2 : inf;
                    }
                }
            } else
            for (int o = wave; o < JO; o += NW) {
                const int j = L.iO[o];
                const double *__restrict__ row = Ct + (size_t)(M + j) * N;
                double az, ap;
                row_dot2<VEC>(row, L.z, L.zm, N, lane, az, ap);
                az = wave_sum(az);
                ap = wave_sum(ap);
                if (lane == 0) L.lin[o] = (ap > tol) ? (rhs[M + j] - az) / ap : inf;
            }
        }
        __syncthreads();
        SUBPHASE(20, tast);
        PCOUNT(26);
        KeyMin ev{inf, 0};
        double Lreg[MPT];
#pragma unroll
        for (int m = 0; m < MPT; ++m) {
            const int k = tid + m * NT;
            Lreg[m] = inf;
            if (k < K) {
                const int i = L.idx[k];
                const double t = L.zm[i], h = L.z[i];
                if (t > tol && ureg[m] < inf) Lreg[m] = (ureg[m] - h) / t;
                else if (t < -tol && dreg[m] > -inf) Lreg[m] = (dreg[m] - h) / t;
                if (Lreg[m] < ev.v) ev.v = Lreg[m];
            }
        }
        for (int o = tid; o < JO; o += NT)
            if (L.lin[o] < ev.v) ev.v = L.lin[o];
        const double L1 = block_min(ev.v, L);
        C.sFlops += 2ll * JO * (N + K);
        if (L1 < 1.0) {  // blocked  (:98-127)
            int firstId = 0x7fffffff;
#pragma unroll
            for (int m = 0; m < MPT; ++m) {
                const int k = tid + m * NT;
                if (k < K) {
                    const int i = L.idx[k];
                    const double t = L.zm[i];
                    double zn = L.z[i] + L1 * t;
                    if (Lreg[m] < inf && !(Lreg[m] - L1 > tol)) {
                        const bool up = t > tol;
                        L.S[i] = up ? SSQP_UP : SSQP_DN;
                        zn = up ? ureg[m] : dreg[m];
                        firstId = min(firstId, i + 1);
                        if (zn != 0.0) {  // B gains a column with a nonzero weight
                            if (evtOK) {
                                const int slot = atomicAdd(&L.ired[2 * NW + 9], 1);
                                if (slot < EVT_MAX) {
                                    L.evti[slot] = i;
                                    L.evtz[slot] = zn;
                                }
                            } else {
                                L.ired[HB_DIRTY] = 1;
                            }
                        }
                    }
                    L.z[i] = zn;
                }
            }
            for (int o = tid; o < JO; o += NT)
                if (L.lin[o] < inf && !(L.lin[o] - L1 > tol)) {
                    L.S[N + L.iO[o]] = SSQP_EO;
                    L.ired[ROWS_DIRTY] = 1;
                    firstId = min(firstId, N + L.iO[o] + 1);
                }
            if (trace) {
                KeyMin f{(double)firstId, 0};
                f = block_keymin(f, L);
                if (tid == 0) *trace = ssqp_trace{K, W, 1, (int)f.v};
            }
            __syncthreads();
            if (evtOK) {  // the caches follow the newly bound variables, by increasing index (deterministic order)
                const int ne = L.ired[2 * NW + 9];
                if (ne > EVT_MAX) {
                    if (tid == 0) L.ired[HB_DIRTY] = 1;
                } else {
                    int last = -1;
                    for (int e = 0; e < ne; ++e) {
                        int best = 0x7fffffff, be = 0;
                        for (int f = 0; f < ne; ++f) {
                            const int iv = L.evti[f];
                            if (iv > last && iv < best) {
                                best = iv;
                                be = f;
                            }
                        }
                        bound_event(L, V, Ct, N, MJ, best, L.evtz[be]);
                        C.sRead += 8ll * N + 64ll * MJ;
                        last = best;
                    }
                }
                __syncthreads();
                if (tid == 0) L.ired[2 * NW + 9] = 0;
            }
            return ACT_CONTINUE;
        }
        // full step: z[F] = alpha  (:130)
#pragma unroll
        for (int m = 0; m < MPT; ++m) {
            const int k = tid + m * NT;
            if (k < K) L.z[L.idx[k]] = areg[m];
        }
    }
    __syncthreads();

    // ---- multipliers: gamma = V[B,F] alpha + V[B,B] zB + q[B] + AB' alphaL  (SSQP.jl:352) ----
    PHASE(C, 9);
    PCOUNT(27);
    if (VEC == 2 && frontDone && !C.dense) {
        // gamma = V[:,F] alpha + AB'alphaL (AXPY over the K free columns and the W kept constraint rows) + hq, the
        // cached hB + q of this pass: the columns of the bound variables are not read a second time
        for (int i = tid; i < N; i += NT) L.zm[i] = (L.pos[i] >= 0) ? L.gam[i] : 0.0;
        for (int t = tid; t < K + W; t += NT) L.perm[t] = (t < K) ? L.idx[t] : (int16_t)(N + (t - K));
        __syncthreads();
        double *stage = (INLDS && (long)NW * N <= C.arenaCap) ? ar : C.garena;
        const AxpyExt ext{Ct, q, L.aL, L.rowsE, L.ra, W};
        stream_axpy<4, 4>(V, N, L.perm, K + W, L.zm, stage, L.gam, ext, L.hq);
        C.sRead += 8ll * N * (K + W);
    } else {
    for (int i = tid; i < N; i += NT) L.zm[i] = (L.pos[i] >= 0) ? L.gam[i] : L.z[i];
    __syncthreads();
    {
        // per-wave partial vectors go through the LDS arena when it has room (the factor is dead), else
        // through the workgroup's global arena
        double *stage = (INLDS && (long)NW * N <= C.arenaCap) ? ar : C.garena;
        // (even N: the constraint-row term AB'*alphaL and q ride along as extra AXPY columns)
        const AxpyExt ext{Ct, q, L.aL, L.rowsE, L.ra, W};
        const int ncols = stream_matvec<VEC>(V, N, L.zm, C.dense, stage, L.gam, L, ext);
        C.sRead += 8ll * N * ncols;
    }
    }
    __syncthreads();
    C.sBytes += 8ll * R * R;
    C.sFlops += 2ll * R * R + 2ll * R * K;

    // ---- KKTchk!  SSQP.jl:136-188 ----
    PHASE(C, 10);
    KeyMin ev{inf, 0x7fffffff};
    for (int i = tid; i < N; i += NT) {
        if (L.pos[i] >= 0) continue;
        double gmm = L.gam[i];
        if (VEC == 1) {  // odd N: the dot-form pass leaves q and AB'*alphaL to be added here
            double s3 = 0.0;
#pragma unroll 4
            for (int w = 0; w < W; ++w) s3 = fma(Ct[(size_t)L.rowsE[L.ra[w]] * N + i], L.aL[w], s3);
            gmm = (gmm + q[i]) + s3;
        }
        const int s = L.S[i];
        if (s == SSQP_UP && gmm > tolG) ev = keymin(ev, KeyMin{-gmm, i});
        else if (s == SSQP_DN && gmm < -tolG) ev = keymin(ev, KeyMin{gmm, i});
    }
    if (JE > 0) {  // multipliers of the active inequalities (:149-171)
        double *Q = fac;               // the factor is dead now
        double *Rm = Q + (long)K * W;  // W x W upper
        double *yv = Rm + W * W;       // W
        if (W < W0) {
            // purged rows: Lda = alphaL' * (AE' \ GE[j,F])  (:158-159); least squares by
            // modified Gram-Schmidt (two passes) on the K x W matrix AE', one wavefront.
            for (int e = tid; e < W * K; e += NT) {
                const int w = e / K, k = e - w * K;
                Q[k + K * w] = Ct[(size_t)L.rowsE[L.ra[w]] * N + L.idx[k]];
            }
            for (int e = tid; e < W * W; e += NT) Rm[e] = 0.0;
            __syncthreads();
            if (wave == 0) {
                for (int c = 0; c < W; ++c) {
                    for (int pass = 0; pass < 2; ++pass)
                        for (int b2 = 0; b2 < c; ++b2) {
                            double s = 0.0;
                            for (int k = lane; k < K; k += 64) s = fma(Q[k + K * b2], Q[k + K * c], s);
                            s = wave_sum(s);
                            for (int k = lane; k < K; k += 64) Q[k + K * c] = fma(-s, Q[k + K * b2], Q[k + K * c]);
                            if (lane == 0) Rm[b2 + W * c] += s;
                        }
                    double s = 0.0;
                    for (int k = lane; k < K; k += 64) s = fma(Q[k + K * c], Q[k + K * c], s);
                    s = sqrt(wave_sum(s));
                    if (lane == 0) Rm[c + W * c] = s;
                    const double rs = (s > 0.0) ? 1.0 / s : 0.0;
                    for (int k = lane; k < K; k += 64) Q[k + K * c] *= rs;
                }
            }
            __syncthreads();
        }
        if (W == W0) {  // every active row kept its own multiplier: one thread per row
            if (tid < JE) {
                const int wrow = M + tid;
                const double Lda = L.aL[wrow];
                if (Lda < -tolG) ev = keymin(ev, KeyMin{Lda, N + (int)L.rowsE[wrow] - M});
            }
        } else
        for (int e = 0; e < JE; ++e) {  // JE is small; uniform loop
            const int wrow = M + e;       // index into rowsE
            int posk = -1;
            if (W == W0) posk = wrow;
            else
                for (int w = 0; w < W; ++w)
                    if (L.ra[w] == wrow) posk = w;
            double Lda;
            if (posk >= 0) {
                Lda = L.aL[posk];
            } else {
                const double *__restrict__ row = Ct + (size_t)L.rowsE[wrow] * N;
                if (wave == 0) {  // yq = Q' gv ; x = R^-1 yq ; Lda = alphaL . x
                    for (int w = 0; w < W; ++w) {
                        double s = 0.0;
                        for (int k = lane; k < K; k += 64) s = fma(Q[k + K * w], row[L.idx[k]], s);
                        s = wave_sum(s);
                        if (lane == 0) yv[w] = s;
                    }
                    if (lane == 0) {
                        for (int c = W - 1; c >= 0; --c) {
                            double s = yv[c];
                            for (int b2 = c + 1; b2 < W; ++b2) s -= Rm[c + W * b2] * yv[b2];
                            yv[c] = (Rm[c + W * c] > 0.0) ? s / Rm[c + W * c] : 0.0;
                        }
                        double s = 0.0;
                        for (int w = 0; w < W; ++w) s = fma(L.aL[w], yv[w], s);
                        L.red[2 * NW - 1] = s;
                    }
                }
                __syncthreads();
                Lda = L.red[2 * NW - 1];
                __syncthreads();
            }
            if (tid == 0 && Lda < -tolG) ev = keymin(ev, KeyMin{Lda, N + (int)L.rowsE[wrow] - M});
            if (tid == 0 && posk < 0 && C.wantMult) L.lin[L.rowsE[wrow]] = Lda;  // (a purged row, as KKTchk! computes it; `lin` is idle here)
        }
    }
    ev = block_keymin(ev, L);
    if (ev.v < inf) {  // release the single tightest one (:175-184)
        if (tid == 0) {
            L.S[ev.ord] = (ev.ord < N) ? SSQP_IN : SSQP_OE;
            if (!evtOK && ev.ord < N && L.z[ev.ord < N ? ev.ord : 0] != 0.0) L.ired[HB_DIRTY] = 1;  // B loses a nonzero column
            if (ev.ord >= N) L.ired[ROWS_DIRTY] = 1;
            if (trace) *trace = ssqp_trace{K, W, 2, ev.ord + 1};
        }
        if (evtOK && ev.ord < N) {  // B loses a column: the caches follow (uniform)
            const double zr = L.z[ev.ord];
            if (zr != 0.0) {  // (hq was last read before the gamma pass's barrier)
                bound_event(L, V, Ct, N, MJ, ev.ord, -zr);
                C.sRead += 8ll * N + 64ll * MJ;
            }
        }
        __syncthreads();
        return ACT_CONTINUE;
    }
    // ---- optimal: when the multipliers of this pass were asked for (alphaL SSQP.jl:351, gamma :352), complete their
    // images in LDS -- `lin` = lambda by row id (purged active inequalities were stored above), `gam` = gamma by variable;
    // solve_one writes them out (it holds the output pointers; carrying them through this function costs registers in
    // every instantiation)
    if (C.wantMult) {
        for (int r = tid; r < MJ; r += NT) {
            int posA = -1;  // position among the active rows
            for (int w = 0; w < W0; ++w)
                if (L.rowsE[w] == r) posA = w;
            int posk = -1;  // ... among the kept rows
            if (posA >= 0) {
                if (W == W0) posk = posA;
                else
                    for (int w = 0; w < W; ++w)
                        if (L.ra[w] == posA) posk = w;
            }
            if (posk >= 0) L.lin[r] = L.aL[posk];
            else if (posA < 0 || r < M) L.lin[r] = 0.0;  // inactive, or a purged equality row (no value in the reference)
        }
        for (int i = tid; i < N; i += NT) {
            double gmm = 0.0;
            if (L.pos[i] < 0) {
                gmm = L.gam[i];
                if (VEC == 1) {
                    double s3 = 0.0;
                    for (int w = 0; w < W; ++w) s3 = fma(Ct[(size_t)L.rowsE[L.ra[w]] * N + i], L.aL[w], s3);
                    gmm = (gmm + q[i]) + s3;
                }
            }
            L.gam[i] = gmm;
        }
    }
    // ---- polishSz!  SSQP.jl:10-32 ----
    PHASE(C, 11);
    for (int i = tid; i < N; i += NT) {
        const int s = L.S[i];
        if (s == SSQP_DN) L.z[i] = dlo[i];
        else if (s == SSQP_UP) L.z[i] = uhi[i];
        else {
            const double zi = L.z[i];
            if (fabs(zi - dlo[i]) < tol) {
                L.z[i] = dlo[i];
                L.S[i] = SSQP_DN;
            } else if (fabs(zi - uhi[i]) < tol) {
                L.z[i] = uhi[i];
                L.S[i] = SSQP_UP;
            }
        }
    }
    __syncthreads();
    for (int j = wave; j < J; j += NW) {
        const double *__restrict__ row = Ct + (size_t)(M + j) * N;
        double az, unused;
        row_dot2<VEC>(row, L.z, L.z, N, lane, az, unused);
        az = wave_sum(az);
        if (lane == 0) L.S[N + j] = (fabs(rhs[M + j] - az) < tol) ? SSQP_EO : SSQP_OE;
    }
    if (trace && tid == 0) *trace = ssqp_trace{K, W, 3, 0};
    C.ret = iter;  // SSQP.jl:374
    return ACT_BREAK;
}

template <int VEC>
__device__ __forceinline__ void solve_one(const SolveParams &P, int prob, const Lds &L, double *garena) {
    const int N = P.N, M = P.M, J = P.J;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ProbCtx C;
    C.N = N; C.M = M; C.J = J; C.MJ = P.MJ;
    C.tol = P.tol; C.tolG = P.tolG;
    // per-problem strides (elements); 0 = one array shared by the whole batch (efficient-frontier style batches)
    C.V = P.V + (size_t)prob * P.sV;
    C.Ct = P.Ct + (size_t)prob * P.sCt;
    C.rhs = P.rhs + (size_t)prob * P.sRhs;
    C.q = P.q + (size_t)prob * P.sq;
    C.dlo = P.d + (size_t)prob * P.sd;
    C.uhi = P.u + (size_t)prob * P.su;
    C.trace = P.trace ? P.trace + (size_t)prob * P.ntrace : nullptr;
    C.ntrace = P.ntrace;
    C.wantMult = P.lamOut != nullptr || P.gamOut != nullptr;
    C.arenaCap = P.arenaCap;
    C.dense = P.denseGamma != 0;
    C.garena = garena;
    C.iter = 0; C.ret = 0; C.det = SSQP_DETAIL_NONE;
    C.sBytes = 0; C.sRead = 0; C.sFlops = 0; C.sK3 = 0; C.maxK = 0; C.pathBits = 0;
    // hand-over from the wavefront kernel (ssqp_wave.hip): continue from its (z, S) at its pass count
    const double *zstart = P.x0 + (size_t)prob * N;
    if (P.resume) {
        C.iter = P.fbIter[prob];
        zstart = P.z + (size_t)prob * N;
        C.pathBits = 32;
    }
    int32_t *Sg = P.S + (size_t)prob * (N + J);
    const double tol = P.tol;
#ifdef SSQP_PHASE_PROFILE
    C.ph_cur = 11;
    C.ph_last = __builtin_amdgcn_s_memtime();
#endif

    for (int i = tid; i < N; i += NT) L.z[i] = zstart[i];
    for (int i = tid; i < N + J; i += NT) L.S[i] = Sg[i];
    for (int i = tid; i < N; i += NT) L.fpos[i] = -1;
    // incremental engine: arena = [scratch (X / AXPY staging / H) | Y | 1/d | factor], capacity RC rows
    C.Kfac = 0;
    C.RC = 0;
    C.yOff = C.rdvOff = C.facOff = 0;
    C.scrCap = P.arenaCap;
    C.facGlobal = false;
    C.hbValid = false;
    C.yValid = 0;
    C.yW0 = -1;
    if (tid == 0) {
        L.ired[HB_DIRTY] = 0;
        L.ired[ROWS_DIRTY] = 1;
        L.ired[2 * NW + 9] = 0;
        L.ired[HQ_CHANGED] = 0;
        L.ired[SHIFT_COUNT] = 0;
    }
    if (P.incremental) {
        int rc = INC_KMAX;
        for (; rc >= 16; rc -= 2) {
            long scr = (long)NW * N + 64;
            const long x = (long)P.MJ * (rc + 2) + 64;
            if (x > scr) scr = x;
            if (scr < 2 * INC_RBM * INC_RBM + 64) scr = 2 * INC_RBM * INC_RBM + 64;  // Schur block + its transposed factor
            const long need = scr + (long)INC_RBM * rc + rc + (long)rc * (rc + 1) / 2 + 8;
            if (need <= P.arenaCap) {
                C.RC = rc;
                C.scrCap = (int)scr;
                C.yOff = (int)scr;
                C.rdvOff = C.yOff + INC_RBM * rc;
                C.facOff = C.rdvOff + rc;
                break;
            }
        }
    }
    __syncthreads();

    for (;;) {
        C.iter += 1;
        if (C.iter > P.maxIter) {  // SSQP.jl:271-274
            C.ret = -C.iter;
            break;
        }
        PHASE(C, 0);
        const int K = compact_free<(VEC == 2) ? 2 : ((VEC == 3) ? 4 : MAXPT)>(L, N);
        ssqp_trace *trace = (C.trace && C.iter <= C.ntrace) ? C.trace + (C.iter - 1) : nullptr;

        if (K == 0) {  // ---------------------------------------- freeK!  SSQP.jl:35-59
            PHASE(C, 12);
            for (int i = tid; i < N; i += NT) L.zm[i] = L.z[i];
            __syncthreads();
            const int ncols = stream_matvec<VEC>(C.V, N, L.zm, C.dense,
                                                 ((long)NW * N <= P.arenaCap) ? L.arena : garena, L.gam, L);
            C.sRead += 8ll * N * ncols + 16ll * N + 4ll * (N + J);
            __syncthreads();
            int flag = 0;
            double pa = 0.0;
            for (int i = tid; i < N; i += NT) {
                const double p = L.gam[i] + C.q[i];
                const int s = L.S[i];
                if ((p >= -tol && s == SSQP_UP) || (p <= tol && s == SSQP_DN)) {
                    flag = 1;
                    pa = fmax(pa, fabs(p));
                    L.pos[i] = -2;  // "to be released"
                }
            }
            C.sBytes += 8ll * N * N + 16ll * N + 4ll * (N + J);
            C.sFlops += 2ll * N * N;
            const int any = block_or(flag, L);
            bool done = !any;
            if (any) {
                const double pm = block_max(pa, L);
                if (pm <= tol) done = true;  // all movable are optimal: statuses restored (:52-55)
            }
            if (done) {
                if (trace && tid == 0) *trace = ssqp_trace{0, 0, 3, 0};
                // (no multipliers exist on this exit of the reference: gamma = V z + q, what freeK! tested; lambda = 0)
                if (C.wantMult) {
                    for (int r = tid; r < P.MJ; r += NT) L.lin[r] = 0.0;
                    for (int i = tid; i < N; i += NT) L.gam[i] = L.gam[i] + C.q[i];
                }
                C.ret = C.iter;  // SSQP.jl:281 (no polishSz! on this exit)
                break;
            }
            for (int i = tid; i < N; i += NT)
                if (L.pos[i] == -2) L.S[i] = SSQP_IN;
            if (tid == 0) L.ired[HB_DIRTY] = 1;
            if (trace && tid == 0) *trace = ssqp_trace{0, 0, 0, 0};
            __syncthreads();
            continue;
        }
        if (K > C.maxK) C.maxK = K;

        // ---- active / inactive inequality lists  (SSQP.jl:288-289) ----
        // (rebuilt only when an inequality changed status since they were formed: ROWS_DIRTY)
        if (wave == 0 && L.ired[ROWS_DIRTY] != 0) {  // J is small: one wavefront, ballot compaction keeps the order
            int nE = M, nO = 0;
            for (int c0 = 0; c0 < J; c0 += 64) {
                const int j = c0 + lane;
                const int s = (j < J) ? L.S[N + j] : -1;
                const unsigned long long mE = __ballot(s == SSQP_EO), mO = __ballot(s == SSQP_OE);
                const unsigned long long lt = (1ull << lane) - 1ull;
                if (s == SSQP_EO) L.rowsE[nE + __popcll(mE & lt)] = (int16_t)(M + j);
                if (s == SSQP_OE) L.iO[nO + __popcll(mO & lt)] = (int16_t)j;
                nE += __popcll(mE);
                nO += __popcll(mO);
            }
            for (int r = lane; r < M; r += 64) L.rowsE[r] = (int16_t)r;
            if (lane == 0) {
                L.ired[2 * NW] = nE;
                L.ired[2 * NW + 1] = nO;
                L.ired[ROWS_DIRTY] = 0;
            }
        }
        __syncthreads();  // pos, idx, zm (zB scattered) and the row lists are published
        const int W0 = L.ired[2 * NW], JO = L.ired[2 * NW + 1];

        // arena: LDS when [X | packed factor + Schur block] fits, else the global scratch
        const long needX = (long)W0 * (K + 1);
        const long R0 = K + W0 + 1;
        long needF = R0 * (R0 + 1) / 2 + R0 + 8;
        const long needLS = (long)K * W0 + (long)W0 * W0 + W0 + 8;
        if (needLS > needF) needF = needLS;
        const bool inLds = (needX <= P.arenaCap) && (needF <= P.arenaCap);
        C.pathBits |= inLds ? 1 : 2;
        // kept-factor engine in the global arena: when K outgrows the LDS factor capacity (up to 256 rows)
        const int RCg = (N < INC_KMAX) ? N : INC_KMAX;
        const bool scratchFits = (long)W0 * (K + 2) + 64 <= P.arenaCap && (long)NW * N + 64 <= P.arenaCap;
        if (P.incremental && !C.facGlobal && K > C.RC && K <= RCg && W0 + 1 <= INC_RBM && scratchFits) {
            // migrate: copy the factor to the global layout (capacity RCg) and continue there
            const int Kf = (C.Kfac > 0 && C.RC > 0) ? C.Kfac : 0;
            double *g0 = garena + ((long)NW * N + 64);          // the front of the global arena stays AXPY staging
            const int yOffG = 0, rdvOffG = INC_RBM * RCg, facOffG = rdvOffG + RCg;
            for (int c = wave; c < Kf; c += NW) {
                const int os = cofs(c, C.RC) - c, od = cofs(c, RCg) - c;
                for (int r = c + lane; r < Kf; r += 64) g0[facOffG + od + r] = L.arena[C.facOff + os + r];
            }
            for (int r = tid; r < Kf; r += NT) g0[rdvOffG + r] = L.arena[C.rdvOff + r];
            if (C.Kfac < 0 || C.RC == 0) {
                for (int i = tid; i < N; i += NT) L.fpos[i] = -1;
                C.Kfac = 0;
            }
            C.facGlobal = true;
            C.yValid = 0;  // (the border columns are not migrated)
            C.RC = RCg;
            C.yOff = (int)((long)NW * N + 64) + yOffG;
            C.rdvOff = (int)((long)NW * N + 64) + rdvOffG;
            C.facOff = (int)((long)NW * N + 64) + facOffG;
            C.scrCap = P.arenaCap;
            C.pathBits |= 8;
            __syncthreads();
        }
        int act;
        if (C.facGlobal && K <= C.RC && W0 + 1 <= INC_RBM && scratchFits) act = iterate_kkt<VEC, true, true>(C, L, L.arena, K, W0, JO);
        else if (inLds) act = iterate_kkt<VEC, true, false>(C, L, L.arena, K, W0, JO);
        else act = iterate_kkt<VEC, false, false>(C, L, garena, K, W0, JO);
        if (act == ACT_BREAK) break;
    }
    __syncthreads();
    PHASE(C, 11);
    if (C.wantMult && C.ret > 0) {  // the last pass left lambda (by row id) in `lin` and gamma (by variable) in `gam`
        if (P.lamOut)
            for (int r = tid; r < P.MJ; r += NT) P.lamOut[(size_t)prob * P.MJ + r] = L.lin[r];
        if (P.gamOut)
            for (int i = tid; i < N; i += NT) P.gamOut[(size_t)prob * N + i] = L.gam[i];
    }
    for (int i = tid; i < N; i += NT) P.z[(size_t)prob * N + i] = L.z[i];
    for (int i = tid; i < N + J; i += NT) Sg[i] = L.S[i];
    if (tid == 0) {
        P.status[prob] = C.ret;
        if (P.detail) P.detail[prob] = C.det;
        if (P.stats) {
            ssqp_stats st;
            st.iters = C.iter > P.maxIter ? P.maxIter : C.iter;
            st.alg_bytes = C.sBytes;
            st.read_bytes = C.sRead;
            st.alg_flops = C.sFlops;
            st.sum_k3 = C.sK3;
            st.max_k = C.maxK;
            st.path = C.pathBits;
            if (P.resume) {  // add what the wavefront kernel counted before the hand-over
                const ssqp_stats s0 = P.stats[prob];
                st.alg_bytes += s0.alg_bytes;
                st.read_bytes += s0.read_bytes;
                st.alg_flops += s0.alg_flops;
                st.sum_k3 += s0.sum_k3;
                st.max_k = st.max_k > s0.max_k ? st.max_k : s0.max_k;
                st.path |= s0.path;
            }
            P.stats[prob] = st;
        }
    }
    PHASE(C, 13);
    __syncthreads();
}

#ifdef SSQP_PHASE_PROFILE
}  // namespace ssqp
extern "C" int ssqp_debug_phases(unsigned long long *out16, int reset) {
    static unsigned long long host[1024 * 32];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(ssqp::g_phase), sizeof(host)) != hipSuccess) return 1;
    for (int k = 0; k < 32; ++k) out16[k] = 0;
    for (int b = 0; b < 1024; ++b)
        for (int k = 0; k < 32; ++k) out16[k] += host[b * 32 + k];
    if (reset) {
        static unsigned long long zero[1024 * 32];
        if (hipMemcpyToSymbol(HIP_SYMBOL(ssqp::g_phase), zero, sizeof(zero)) != hipSuccess) return 1;
    }
    return 0;
}
namespace ssqp {
#endif

// WPS = waves per SIMD the register allocation must allow = workgroups per CU (one wave per SIMD each)
template <int VEC, int WPS>
__global__ __launch_bounds__(NT, WPS) void ssqp_solve_kernel(SolveParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Lds L;
    {
        const LdsLayout lay = lds_layout(P.N, P.M, P.J, P.arenaCap);
        double *d0 = reinterpret_cast<double *>(smem);
        L.z = d0 + lay.z;
        L.zm = d0 + lay.zm;
        L.gam = d0 + lay.gam;
        L.hq = d0 + lay.hq;
        L.arena = d0 + lay.arena;
        L.bE = d0 + lay.bE;
        L.aL = d0 + lay.aL;
        L.tv = d0 + lay.tv;
        L.dcol = d0 + lay.dcol;
        L.lin = d0 + lay.lin;
        L.bEall = d0 + lay.bEall;
        L.red = d0 + lay.red;
        L.S = reinterpret_cast<int32_t *>(smem + lay.S_bytes);
        L.ired = reinterpret_cast<int *>(smem + lay.ired_bytes);
        L.pos = reinterpret_cast<int16_t *>(smem + lay.pos_bytes);
        L.idx = reinterpret_cast<int16_t *>(smem + lay.idx_bytes);
        L.perm = reinterpret_cast<int16_t *>(smem + lay.perm_bytes);
        L.rowsE = reinterpret_cast<int16_t *>(smem + lay.rowsE_bytes);
        L.ra = reinterpret_cast<int16_t *>(smem + lay.ra_bytes);
        L.iO = reinterpret_cast<int16_t *>(smem + lay.iO_bytes);
        L.fpos = reinterpret_cast<int16_t *>(smem + lay.fpos_bytes);
        L.ordl = reinterpret_cast<int16_t *>(smem + lay.ordl_bytes);
        L.ytag = reinterpret_cast<int16_t *>(smem + lay.ytag_bytes);
        L.evtz = reinterpret_cast<double *>(smem + lay.evt_bytes);
        L.evti = reinterpret_cast<int *>(smem + lay.evt_bytes + 16 * 8);
    }
    double *garena = P.gscratch + (size_t)blockIdx.x * P.gscratchStride;
    for (;;) {
        if (threadIdx.x == 0) L.ired[2 * NW + 3] = (int)atomicAdd(P.queue, 1u);
        __syncthreads();
        int prob = L.ired[2 * NW + 3];
        __syncthreads();
        if (P.resume) {  // the QPs the wavefront kernel handed over
            if (prob >= (int)*P.fbCount) break;
            prob = P.fbList[prob];
        } else if (prob >= P.nprob) {
            break;
        }
        solve_one<VEC>(P, prob, L, garena);
    }
}

// Ct[p][r][i] = [A;G][r, i]: constraint rows made contiguous; rhs = [b; g].  nct / nrhs = number of distinct
// Ct / rhs images (1 when A and G, resp. b and g, are shared by the batch); sA.. = element strides (0 = shared).
__global__ void ssqp_prep_kernel(int nct, int nrhs, int N, int M, int J, const double *__restrict__ A,
                                 const double *__restrict__ G, const double *__restrict__ b,
                                 const double *__restrict__ g, size_t sA, size_t sG, size_t sb, size_t sg,
                                 double *__restrict__ Ct, double *__restrict__ rhs) {
    const int MJ = M + J;
    const size_t total = (size_t)nct * MJ * N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const size_t p = e / ((size_t)MJ * N);
        const size_t rem = e - p * (size_t)MJ * N;
        const int r = (int)(rem / N), i = (int)(rem - (size_t)r * N);
        Ct[e] = (r < M) ? A[p * sA + (size_t)i * M + r] : G[p * sG + (size_t)i * J + (r - M)];
    }
    const size_t tr = (size_t)nrhs * MJ;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tr; e += (size_t)gridDim.x * blockDim.x) {
        const size_t p = e / MJ;
        const int r = (int)(e - p * MJ);
        rhs[e] = (r < M) ? b[p * sb + r] : g[p * sg + (r - M)];
    }
}

void launch_prep(int nct, int nrhs, int N, int M, int J, const double *A, const double *G, const double *b,
                 const double *g, size_t sA, size_t sG, size_t sb, size_t sg, double *Ct, double *rhs,
                 hipStream_t stream) {
    if (M + J == 0 || (nct == 0 && nrhs == 0)) return;
    const size_t total = (size_t)(nct > nrhs ? nct : nrhs) * (M + J) * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(ssqp_prep_kernel, dim3(blocks), dim3(256), 0, stream, nct, nrhs, N, M, J, A, G, b, g, sA, sG, sb,
                       sg, Ct, rhs);
}

// ---------------------------------------------------------------- synthetic V on the device
// Bit-identical to generate_one() in ssqp_host.cpp: X[t,i] = u01(stream X, t + T*i) - 1/2,
// V[i,j] = (sum over t in increasing order of X[t,i]*X[t,j]) / T + delta*(i==j), products and sums rounded
// separately (no FMA contraction).  One thread per (i <= j) pair mirrors its result.
__device__ __forceinline__ unsigned long long gen_mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void ssqp_genV_kernel(int nprob, int N, int T, double delta,
                                                        unsigned long long seed0, double *__restrict__ V) {
#pragma clang fp contract(off)
    constexpr unsigned long long GOLDEN = 0x9E3779B97F4A7C15ull;
    constexpr int TILE = 16, TB = 64;
    __shared__ double xi[TB][TILE + 1], xj[TB][TILE + 1];
    const int tilesPerDim = (N + TILE - 1) / TILE;
    const int ntile = tilesPerDim * (tilesPerDim + 1) / 2;
    const int p = blockIdx.x / ntile;
    if (p >= nprob) return;
    int tIdx = blockIdx.x - p * ntile, tj = 0;
    while (tIdx > tj) {  // upper-triangular tile index -> (ti <= tj)
        tIdx -= tj + 1;
        ++tj;
    }
    const int ti = tIdx;
    const unsigned long long seed = seed0 + (unsigned long long)p;
    const unsigned long long base = gen_mix64(gen_mix64(seed + GOLDEN) ^ (1ull * 0xD1B54A32D192ED03ull + GOLDEN));
    const int li = threadIdx.x & 15, lj = threadIdx.x >> 4;
    const int i = ti * TILE + li, j = tj * TILE + lj;
    double acc = 0.0;
    for (int t0 = 0; t0 < T; t0 += TB) {
        for (int e = threadIdx.x; e < TB * TILE; e += 256) {
            const int tt = e / TILE, c = e - tt * TILE;
            const int t = t0 + tt;
            const int ci = ti * TILE + c, cj = tj * TILE + c;
            double a = 0.0, b = 0.0;
            if (t < T && ci < N)
                a = (double)(gen_mix64(base + ((unsigned long long)t + (unsigned long long)T * ci + 1ull) * GOLDEN) >> 11) * 0x1.0p-53 - 0.5;
            if (t < T && cj < N)
                b = (double)(gen_mix64(base + ((unsigned long long)t + (unsigned long long)T * cj + 1ull) * GOLDEN) >> 11) * 0x1.0p-53 - 0.5;
            xi[tt][c] = a;
            xj[tt][c] = b;
        }
        __syncthreads();
        const int tb = (T - t0 < TB) ? T - t0 : TB;
        for (int tt = 0; tt < tb; ++tt) {
            const double prod = xi[tt][li] * xj[tt][lj];
            acc = acc + prod;
        }
        __syncthreads();
    }
    if (i < N && j < N && i <= j) {
        double v = acc / (double)T;
        if (i == j) v = v + delta;
        double *Vp = V + (size_t)p * N * N;
        Vp[(size_t)j * N + i] = v;
        Vp[(size_t)i * N + j] = v;
    }
}

hipError_t launch_genV(int nprob, int N, int T, double delta, unsigned long long seed0, double *V, hipStream_t stream) {
    const int tiles = (N + 15) / 16;
    const long blocks = (long)nprob * tiles * (tiles + 1) / 2;
    hipLaunchKernelGGL(ssqp_genV_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, nprob, N, T, delta, seed0, V);
    return hipGetLastError();
}

template <int VEC, int WPS>
static hipError_t launch_one(const SolveParams &P, int grid, size_t ldsBytes, hipStream_t stream) {
    static unsigned long long ldsSet = 0ull;  // (one per instantiation)
    hipError_t e = allow_full_lds(reinterpret_cast<const void *>(&ssqp_solve_kernel<VEC, WPS>), &ldsSet);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((ssqp_solve_kernel<VEC, WPS>), dim3(grid), dim3(NT), ldsBytes, stream, P);
    return hipGetLastError();
}


hipError_t launch_solve(const SolveParams &P, int grid, size_t ldsBytes, int wgPerCU, hipStream_t stream) {
    // load/accumulate mode: see stream_matvec
    const int mode = (P.N & 1) ? 1 : (P.N <= 512 ? 2 : (P.N <= 1024 ? 3 : 4));
    // (register budget of two workgroups per CU; three were measured slower -- 168 VGPRs spill -- and are not built)
    (void)wgPerCU;
    switch (mode) {
        case 1: return launch_one<1, 2>(P, grid, ldsBytes, stream);
        case 2: return launch_one<2, 2>(P, grid, ldsBytes, stream);
        case 3: return launch_one<3, 1>(P, grid, ldsBytes, stream);  // N > 512: one workgroup per CU, up to 512 VGPRs
        default: return launch_one<4, 1>(P, grid, ldsBytes, stream);
    }
}

}  // namespace ssqp
