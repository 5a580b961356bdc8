// ssqp_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the active-set inner loop.
//
// One 512-thread workgroup owns one QP and runs the WHOLE solveQP(Q,S,x0) loop
// (reference: src/SSQP.jl:237-377) in-kernel: no host round trip per iteration.
// Workgroups are persistent and pull problem ids from a device counter.
//
// Per loop pass (names follow the reference):
//   compaction      F = (S .== IN), index lists               SSQP.jl:276-289
//   E-row sweep     bE = [b; g[Eg]] - AB*zB, X = [AE bE]      SSQP.jl:290-295
//   rank filter     getRowsGJr(X, tol)                        utils.jl:49-86
//   pass 1          stream V[:,F]: c = V[B,F]'zB + q[F] and the K x K gather
//                   V[F,F] straight into the LDS factor        SSQP.jl:322-324
//   KKT solve       bordered LDL' of [V_FF AE' c] in LDS, Schur system
//                   (AE V_FF^-1 AE') lambda = bE + AE V_FF^-1 c, one back
//                   substitution for alpha                     SSQP.jl:325-332
//   aStep!          ratio test as workgroup min-reduction      SSQP.jl:61-134
//   pass 2          stream V[:,B]: gamma                       SSQP.jl:351-352
//   KKTchk!         (value, order) argmin                      SSQP.jl:136-188
//   polishSz!                                                  SSQP.jl:10-32
//   freeK!          K == 0 pass                                SSQP.jl:35-59
//
// The reference forms inv(cholesky(V[F,F])) and inv(cholesky(C)) explicitly;
// this kernel factors and solves (same KKT system).
// Status decisions are threshold tests far above rounding noise, so S matches
// bit for bit away from exact ties; z/lambda agree to ~1e-13 relative.
//
// HBM traffic per pass is the two column sweeps of V (K columns, then N-K
// columns, each column read once, 16 B per lane, a full 1 KiB per wave
// instruction); everything else lives in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssqp_hip.h"
#include "ssqp_internal.h"

namespace ssqp {

// orders the LDS/global accesses of the lanes of ONE wavefront (the wave runs in
// lockstep; this only stops the compiler from moving accesses across it and
// waits for outstanding ones)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// x - d*y and x/d with the reference's two roundings (no FMA contraction)
__device__ __forceinline__ double sub_mul_nc(double x, double d, double y) {
#pragma clang fp contract(off)
    const double t = d * y;
    return x - t;
}

// ---------------------------------------------------------------- reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

struct KeyMin {  // minimum value, ties -> smallest order
    double v;
    int ord;
};
__device__ __forceinline__ KeyMin keymin(KeyMin a, KeyMin b) {
    return (b.v < a.v || (b.v == a.v && b.ord < a.ord)) ? b : a;
}
__device__ __forceinline__ KeyMin wave_keymin(KeyMin a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        KeyMin b;
        b.v = __shfl_xor(a.v, o, 64);
        b.ord = __shfl_xor(a.ord, o, 64);
        a = keymin(a, b);
    }
    return a;
}

struct Lds {
    double *z, *zm, *gam, *arena;
    double *bE, *aL, *tv, *dcol, *lin;  // MJ+1 each
    double *red;                         // 2*NW
    int32_t *S;
    int *ired;                           // 2*NW + 8
    int16_t *pos, *idx, *perm, *rowsE, *ra, *iO;
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
    if (VEC == 2) {
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
    if (VEC == 2) {
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

// out[col] = V[:,col] . w   for the n columns listed in cols (stride cstep:
// +1 walks the free list from the front, -1 walks the bound list from the back)
template <int VEC>
__device__ __forceinline__ void stream_cols(const double *__restrict__ V, int N, const int16_t *cols,
                                            int cstep, int n, const double *w, double *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave * 2; t < n; t += NW * 2) {
        const bool two = (t + 1 < n);
        const int j0 = cols[t * cstep];
        const int j1 = two ? cols[(t + 1) * cstep] : j0;
        double a0, a1;
        dot2_cols<VEC>(V + (size_t)j0 * N, V + (size_t)j1 * N, w, N, lane, a0, a1);
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        if (lane == 0) {
            out[j0] = a0;
            if (two) out[j1] = a1;
        }
    }
}
template <int VEC>
__device__ __forceinline__ void stream_all_cols(const double *__restrict__ V, int N, const double *w,
                                                double *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave * 2; t < N; t += NW * 2) {
        const bool two = (t + 1 < N);
        const int j0 = t, j1 = two ? t + 1 : t;
        double a0, a1;
        dot2_cols<VEC>(V + (size_t)j0 * N, V + (size_t)j1 * N, w, N, lane, a0, a1);
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        if (lane == 0) {
            out[j0] = a0;
            if (two) out[j1] = a1;
        }
    }
}

// ------------------------------------------------------------- compaction
// pos[i] = rank of i among the free variables or -1; idx[0..K) = free indices
// (increasing, like findall, SSQP.jl:276); idx[N-1-r] = r-th bound index.
__device__ __forceinline__ int compact_free(const Lds &L, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int c0 = 0; c0 < N; c0 += NT) {
        const int i = c0 + threadIdx.x;
        const bool f = (i < N) && (L.S[i] == SSQP_IN);
        const unsigned long long m = __ballot(f);
        const int lp = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) L.ired[wave] = __popcll(m);
        __syncthreads();
        int wb = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int c = L.ired[w];
            if (w < wave) wb += c;
            tot += c;
        }
        if (i < N) {
            if (f) {
                const int p = base + wb + lp;
                L.pos[i] = (int16_t)p;
                L.idx[p] = (int16_t)i;
            } else {
                L.pos[i] = -1;
                const int r = i - (base + wb + lp);  // rank among bound variables
                L.idx[N - 1 - r] = (int16_t)i;
            }
        }
        base += tot;
        __syncthreads();
    }
    return base;
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

// --------------------------------------------------- bordered LDL' in the arena
// fac: packed lower triangle of [V_FF; AE; c'] (R = K+W+1 rows, K columns).
// After the call column j holds the unscaled eliminated entries a(i,j), rd[j] =
// 1/d_j; unit-lower L(i,j) = a(i,j)*rd[j], border rows = L^-1 [AE' c].
// Returns false when a pivot is not > 0 (the reference's cholesky throws).
__device__ __forceinline__ bool bordered_ldl(double *fac, double *rd, int K, int R, const Lds &L) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool ok = true;
    for (int j = 0; j < K; ++j) {
        __syncthreads();
        const int oj = coloff(j, R);
        const double d = fac[oj];
        if (!(d > 0.0)) {
            ok = false;
            break;
        }
        const double r = 1.0 / d;
        if (threadIdx.x == 0) rd[j] = r;
        for (int k = j + 1 + wave; k < K; k += NW) {
            const double f = fac[oj + k - j] * r;
            const int ok_ = coloff(k, R);
            for (int i = k + lane; i < R; i += 64) fac[ok_ + i - k] = fma(-f, fac[oj + i - j], fac[ok_ + i - k]);
        }
    }
    __syncthreads();
    return ok;
}

// Solve the unit upper system L' x = v in place (v in vk[0..K)), one wave.
__device__ __forceinline__ void back_substitute(const double *fac, const double *rd, double *vk, int K, int R) {
    const int lane = threadIdx.x & 63;
    for (int j = K - 1; j > 0; --j) {
        const double xj = vk[j];
        for (int i = lane; i < j; i += 64) {
            const double lij = fac[coloff(i, R) + j - i] * rd[i];
            vk[i] = fma(-lij, xj, vk[i]);
        }
        wave_sync();
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
    // results / accounting
    int64_t iter, ret;
    int32_t det;
    int64_t sBytes, sFlops, sK3;
    int maxK, pathBits;
};

enum { ACT_CONTINUE = 0, ACT_BREAK = 1 };

// One pass of the loop for K > 0, from the E-row sweep to the status switch.
// INLDS selects where the arena (X, packed factor, Schur block) lives: the
// instantiation with INLDS=true only ever sees LDS pointers.
template <int VEC, bool INLDS>
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
    const int64_t iter = C.iter;
    ssqp_trace *trace = (C.trace && iter <= C.ntrace) ? C.trace + (iter - 1) : nullptr;

    // ---- E-row sweep: bE and X = [AE bE]   (SSQP.jl:290-295) ----
    {
        double *X = ar;
        for (int w = wave; w < W0; w += NW) {
            const int r = L.rowsE[w];
            const double *__restrict__ row = Ct + (size_t)r * N;
            double acc = 0.0;
            for (int i = lane; i < N; i += 64) {
                const double v = row[i];
                acc = fma(v, L.zm[i], acc);
                const int p = L.pos[i];
                if (p >= 0) X[w + W0 * p] = v;
            }
            acc = wave_sum(acc);
            if (lane == 0) {
                const double be = rhs[r] - acc;
                L.bE[w] = be;
                X[w + W0 * K] = be;
            }
        }
    }
    __syncthreads();
    // ---- rank filter  (SSQP.jl:310-319) ----
    int W = W0;
    if (W0 > 0) W = rank_filter(ar, W0, K + 1, tol, L);
    if (W < W0) {
        double v = 0.0;
        if (tid < W) v = L.bE[L.ra[tid]];
        __syncthreads();
        if (tid < W) L.bE[tid] = v;
    } else {
        for (int w = tid; w < W0; w += NT) L.ra[w] = (int16_t)w;
    }
    __syncthreads();

    // ---- factor assembly: pass 1 over V[:,F] + border rows ----
    const int Rr = K + W + 1;
    double *fac = ar;
    double *rd = ar + coloff(K, Rr);
    double *H = rd + K;  // W x W, column-major, lower part used
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
    // (barrier at the top of bordered_ldl)
    if (!bordered_ldl(fac, rd, K, Rr, L)) {
        C.ret = -1;
        C.det = SSQP_DETAIL_POSDEF_V;
        return ACT_BREAK;
    }
    // ---- Schur system: H = AE V^-1 AE', t = AE V^-1 c ----
    for (int e = wave; e < W * (W + 1) / 2 + W; e += NW) {
        int a, b;  // border rows a >= b; a == W is the c row
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
        double acc = 0.0;
        for (int j = lane; j < K; j += 64) {
            const int oj = coloff(j, Rr);
            acc = fma(fac[oj + K + a - j] * rd[j], fac[oj + K + b - j], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            if (e < W) L.tv[b] = acc;
            else H[a + W * b] = acc;
        }
    }
    __syncthreads();
    // lambda: H lam = bE + t ; alphaL = -lam  (SSQP.jl:351, algebraically)
    if (wave == 0) {
        bool okH = true;
        for (int c = 0; c < W; ++c) {  // in-place LDL' of H, lanes over rows
            const double dcc = H[c + W * c];
            if (!(dcc > 0.0)) {
                okH = false;
                break;
            }
            const double rc = 1.0 / dcc;
            for (int k2 = c + 1; k2 < W; ++k2) {
                const double f = H[k2 + W * c] * rc;
                for (int i = k2 + lane; i < W; i += 64) H[i + W * k2] = fma(-f, H[i + W * c], H[i + W * k2]);
            }
            wave_sync();
        }
        if (lane == 0) {
            L.ired[2 * NW + 2] = okH ? 1 : 0;
            if (okH) {  // forward (unit lower), diagonal, backward: W is small
                for (int w = 0; w < W; ++w) L.aL[w] = L.bE[w] + L.tv[w];
                for (int c = 0; c < W; ++c) {
                    const double xc = L.aL[c];
                    const double rc = 1.0 / H[c + W * c];
                    for (int i = c + 1; i < W; ++i) L.aL[i] = fma(-H[i + W * c] * rc, xc, L.aL[i]);
                }
                for (int c = 0; c < W; ++c) L.aL[c] = L.aL[c] / H[c + W * c];
                for (int c = W - 1; c >= 0; --c) {
                    double xc = L.aL[c];
                    const double rc = 1.0 / H[c + W * c];
                    for (int i = c + 1; i < W; ++i) xc = fma(-H[i + W * c] * rc, L.aL[i], xc);
                    L.aL[c] = xc;
                }
                for (int w = 0; w < W; ++w) L.aL[w] = -L.aL[w];
            }
        }
    }
    __syncthreads();
    if (!L.ired[2 * NW + 2]) {
        C.ret = -1;
        C.det = SSQP_DETAIL_POSDEF_C;
        return ACT_BREAK;
    }
    // v = D^-1 (Y_A alphaL + y_c); alpha = -L'^-1 v
    double *vk = L.gam;
    for (int j = tid; j < K; j += NT) {
        const int oj = coloff(j, Rr);
        double s = fac[oj + K + W - j];
        for (int w = 0; w < W; ++w) s = fma(fac[oj + K + w - j], L.aL[w], s);
        vk[j] = s * rd[j];
    }
    __syncthreads();
    if (wave == 0) back_substitute(fac, rd, vk, K, Rr);
    __syncthreads();
    // ---- alpha (scattered into gam), p (scattered into zm) ----
    double areg[MAXPT];
#pragma unroll
    for (int m = 0; m < MAXPT; ++m) {
        const int k = tid + m * NT;
        areg[m] = (k < K) ? -vk[k] : 0.0;
    }
    __syncthreads();
    double pa = 0.0;
    int pnan = 0;
    for (int i = tid; i < N; i += NT)
        if (L.pos[i] < 0) L.zm[i] = 0.0;
#pragma unroll
    for (int m = 0; m < MAXPT; ++m) {
        const int k = tid + m * NT;
        if (k < K) {
            const int i = L.idx[k];
            const double p = areg[m] - L.z[i];
            L.gam[i] = areg[m];
            L.zm[i] = p;
            if (p != p) pnan = 1;
            pa = fmax(pa, fabs(p));
        }
    }
    const double pinf = block_max(pa, L);  // its barriers also order the writes above
    const int anyNan = block_or(pnan, L);

    // per-pass accounting (the R^2 terms only when gamma is formed)
    {
        const long long k = K, r = R, w = W;
        C.sBytes += 8ll * (k * k + r * k) + 8ll * MJ * N + 48ll * N + 4ll * (N + J);
        C.sFlops += k * k * k + 4 * k * k * w + 2 * k * k + 2 * r * k + 2ll * W0 * r + w * w * w;
        C.sK3 += k * k * k;
    }

    if (pinf > tolG && !anyNan) {  // ------------------------ aStep!  SSQP.jl:61-134
        // inactive inequalities: zo = g - G z, po = G[:,F] p   (:78-89)
        for (int o = wave; o < JO; o += NW) {
            const int j = L.iO[o];
            const double *__restrict__ row = Ct + (size_t)(M + j) * N;
            double az = 0.0, ap = 0.0;
            for (int i = lane; i < N; i += 64) {
                const double v = row[i];
                az = fma(v, L.z[i], az);
                ap = fma(v, L.zm[i], ap);
            }
            az = wave_sum(az);
            ap = wave_sum(ap);
            if (lane == 0) L.lin[o] = (ap > tol) ? (rhs[M + j] - az) / ap : inf;
        }
        __syncthreads();
        KeyMin ev{inf, 0};
        double Lreg[MAXPT];
#pragma unroll
        for (int m = 0; m < MAXPT; ++m) {
            const int k = tid + m * NT;
            Lreg[m] = inf;
            if (k < K) {
                const int i = L.idx[k];
                const double t = L.zm[i], h = L.z[i];
                if (t > tol && uhi[i] < inf) Lreg[m] = (uhi[i] - h) / t;
                else if (t < -tol && dlo[i] > -inf) Lreg[m] = (dlo[i] - h) / t;
                if (Lreg[m] < ev.v) ev.v = Lreg[m];
            }
        }
        for (int o = tid; o < JO; o += NT)
            if (L.lin[o] < ev.v) ev.v = L.lin[o];
        const double L1 = block_keymin(ev, L).v;
        C.sFlops += 2ll * JO * (N + K);
        if (L1 < 1.0) {  // blocked  (:98-127)
            int firstId = 0x7fffffff;
#pragma unroll
            for (int m = 0; m < MAXPT; ++m) {
                const int k = tid + m * NT;
                if (k < K) {
                    const int i = L.idx[k];
                    const double t = L.zm[i];
                    double zn = L.z[i] + L1 * t;
                    if (Lreg[m] < inf && !(Lreg[m] - L1 > tol)) {
                        const bool up = t > tol;
                        L.S[i] = up ? SSQP_UP : SSQP_DN;
                        zn = up ? uhi[i] : dlo[i];
                        firstId = min(firstId, i + 1);
                    }
                    L.z[i] = zn;
                }
            }
            for (int o = tid; o < JO; o += NT)
                if (L.lin[o] < inf && !(L.lin[o] - L1 > tol)) {
                    L.S[N + L.iO[o]] = SSQP_EO;
                    firstId = min(firstId, N + L.iO[o] + 1);
                }
            if (trace) {
                KeyMin f{(double)firstId, 0};
                f = block_keymin(f, L);
                if (tid == 0) *trace = ssqp_trace{K, W, 1, (int)f.v};
            }
            __syncthreads();
            return ACT_CONTINUE;
        }
        // full step: z[F] = alpha  (:130)
#pragma unroll
        for (int m = 0; m < MAXPT; ++m) {
            const int k = tid + m * NT;
            if (k < K) L.z[L.idx[k]] = areg[m];
        }
    }
    __syncthreads();

    // ---- multipliers: gamma = V[B,F] alpha + V[B,B] zB + q[B] + AB' alphaL  (SSQP.jl:352) ----
    for (int i = tid; i < N; i += NT) L.zm[i] = (L.pos[i] >= 0) ? L.gam[i] : L.z[i];
    __syncthreads();
    stream_cols<VEC>(V, N, L.idx + (N - 1), -1, R, L.zm, L.gam);
    __syncthreads();
    C.sBytes += 8ll * R * R;
    C.sFlops += 2ll * R * R + 2ll * R * K;

    // ---- KKTchk!  SSQP.jl:136-188 ----
    KeyMin ev{inf, 0x7fffffff};
    for (int i = tid; i < N; i += NT) {
        if (L.pos[i] >= 0) continue;
        double gmm = L.gam[i] + q[i];
        double s3 = 0.0;
        for (int w = 0; w < W; ++w) s3 = fma(Ct[(size_t)L.rowsE[L.ra[w]] * N + i], L.aL[w], s3);
        gmm += s3;
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
        }
    }
    ev = block_keymin(ev, L);
    if (ev.v < inf) {  // release the single tightest one (:175-184)
        if (tid == 0) {
            L.S[ev.ord] = (ev.ord < N) ? SSQP_IN : SSQP_OE;
            if (trace) *trace = ssqp_trace{K, W, 2, ev.ord + 1};
        }
        __syncthreads();
        return ACT_CONTINUE;
    }
    // ---- optimal: polishSz!  SSQP.jl:10-32 ----
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
        double az = 0.0;
        for (int i = lane; i < N; i += 64) az = fma(row[i], L.z[i], az);
        az = wave_sum(az);
        if (lane == 0) L.S[N + j] = (fabs(rhs[M + j] - az) < tol) ? SSQP_EO : SSQP_OE;
    }
    if (trace && tid == 0) *trace = ssqp_trace{K, W, 3, 0};
    C.ret = iter;  // SSQP.jl:374
    return ACT_BREAK;
}

template <int VEC>
__device__ void solve_one(const SolveParams &P, int prob, const Lds &L, double *garena) {
    const int N = P.N, M = P.M, J = P.J;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ProbCtx C;
    C.N = N; C.M = M; C.J = J; C.MJ = P.MJ;
    C.tol = P.tol; C.tolG = P.tolG;
    C.V = P.V + (size_t)prob * N * N;
    C.Ct = P.Ct + (size_t)prob * N * P.MJ;
    C.rhs = P.rhs + (size_t)prob * P.MJ;
    C.q = P.q + (size_t)prob * N;
    C.dlo = P.d + (size_t)prob * N;
    C.uhi = P.u + (size_t)prob * N;
    C.trace = P.trace ? P.trace + (size_t)prob * P.ntrace : nullptr;
    C.ntrace = P.ntrace;
    C.iter = 0; C.ret = 0; C.det = SSQP_DETAIL_NONE;
    C.sBytes = 0; C.sFlops = 0; C.sK3 = 0; C.maxK = 0; C.pathBits = 0;
    int32_t *Sg = P.S + (size_t)prob * (N + J);
    const double tol = P.tol;

    for (int i = tid; i < N; i += NT) L.z[i] = P.x0[(size_t)prob * N + i];
    for (int i = tid; i < N + J; i += NT) L.S[i] = Sg[i];
    __syncthreads();

    for (;;) {
        C.iter += 1;
        if (C.iter > P.maxIter) {  // SSQP.jl:271-274
            C.ret = -C.iter;
            break;
        }
        const int K = compact_free(L, N);
        ssqp_trace *trace = (C.trace && C.iter <= C.ntrace) ? C.trace + (C.iter - 1) : nullptr;

        if (K == 0) {  // ---------------------------------------- freeK!  SSQP.jl:35-59
            for (int i = tid; i < N; i += NT) L.zm[i] = L.z[i];
            __syncthreads();
            stream_all_cols<VEC>(C.V, N, L.zm, L.gam);
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
                C.ret = C.iter;  // SSQP.jl:281 (no polishSz! on this exit)
                break;
            }
            for (int i = tid; i < N; i += NT)
                if (L.pos[i] == -2) L.S[i] = SSQP_IN;
            if (trace && tid == 0) *trace = ssqp_trace{0, 0, 0, 0};
            __syncthreads();
            continue;
        }
        if (K > C.maxK) C.maxK = K;

        // ---- active / inactive inequality lists  (SSQP.jl:288-289) ----
        if (wave == 0) {  // J is small: one wavefront, ballot compaction keeps the order
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
            }
        }
        for (int i = tid; i < N; i += NT) L.zm[i] = (L.pos[i] >= 0) ? 0.0 : L.z[i];  // zB scattered
        __syncthreads();
        const int W0 = L.ired[2 * NW], JO = L.ired[2 * NW + 1];

        // arena: LDS when [X | packed factor + Schur block] fits, else the global scratch
        const long needX = (long)W0 * (K + 1);
        const long needF = (long)coloff(K, K + W0 + 1) + K + (long)W0 * W0 + W0 + 8;
        const bool inLds = (needX <= P.arenaCap) && (needF <= P.arenaCap);
        C.pathBits |= inLds ? 1 : 2;
        int act;
        if (inLds) act = iterate_kkt<VEC, true>(C, L, L.arena, K, W0, JO);
        else act = iterate_kkt<VEC, false>(C, L, garena, K, W0, JO);
        if (act == ACT_BREAK) break;
    }
    __syncthreads();
    for (int i = tid; i < N; i += NT) P.z[(size_t)prob * N + i] = L.z[i];
    for (int i = tid; i < N + J; i += NT) Sg[i] = L.S[i];
    if (tid == 0) {
        P.status[prob] = C.ret;
        if (P.detail) P.detail[prob] = C.det;
        if (P.stats) {
            ssqp_stats st;
            st.iters = C.iter > P.maxIter ? P.maxIter : C.iter;
            st.alg_bytes = C.sBytes;
            st.alg_flops = C.sFlops;
            st.sum_k3 = C.sK3;
            st.max_k = C.maxK;
            st.path = C.pathBits;
            P.stats[prob] = st;
        }
    }
    __syncthreads();
}

template <int VEC>
__global__ __launch_bounds__(NT) void ssqp_solve_kernel(SolveParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Lds L;
    {
        const LdsLayout lay = lds_layout(P.N, P.M, P.J, P.arenaCap);
        double *d0 = reinterpret_cast<double *>(smem);
        L.z = d0 + lay.z;
        L.zm = d0 + lay.zm;
        L.gam = d0 + lay.gam;
        L.arena = d0 + lay.arena;
        L.bE = d0 + lay.bE;
        L.aL = d0 + lay.aL;
        L.tv = d0 + lay.tv;
        L.dcol = d0 + lay.dcol;
        L.lin = d0 + lay.lin;
        L.red = d0 + lay.red;
        L.S = reinterpret_cast<int32_t *>(smem + lay.S_bytes);
        L.ired = reinterpret_cast<int *>(smem + lay.ired_bytes);
        L.pos = reinterpret_cast<int16_t *>(smem + lay.pos_bytes);
        L.idx = reinterpret_cast<int16_t *>(smem + lay.idx_bytes);
        L.perm = reinterpret_cast<int16_t *>(smem + lay.perm_bytes);
        L.rowsE = reinterpret_cast<int16_t *>(smem + lay.rowsE_bytes);
        L.ra = reinterpret_cast<int16_t *>(smem + lay.ra_bytes);
        L.iO = reinterpret_cast<int16_t *>(smem + lay.iO_bytes);
    }
    double *garena = P.gscratch + (size_t)blockIdx.x * P.gscratchStride;
    for (;;) {
        if (threadIdx.x == 0) L.ired[2 * NW + 3] = (int)atomicAdd(P.queue, 1u);
        __syncthreads();
        const int prob = L.ired[2 * NW + 3];
        __syncthreads();
        if (prob >= P.nprob) break;
        solve_one<VEC>(P, prob, L, garena);
    }
}

// Ct[p][r][i] = [A;G][r, i]: constraint rows made contiguous; rhs = [b; g]
__global__ void ssqp_prep_kernel(int nprob, int N, int M, int J, const double *__restrict__ A,
                                 const double *__restrict__ G, const double *__restrict__ b,
                                 const double *__restrict__ g, double *__restrict__ Ct,
                                 double *__restrict__ rhs) {
    const int MJ = M + J;
    const size_t total = (size_t)nprob * MJ * N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const size_t p = e / ((size_t)MJ * N);
        const size_t rem = e - p * (size_t)MJ * N;
        const int r = (int)(rem / N), i = (int)(rem - (size_t)r * N);
        Ct[e] = (r < M) ? A[p * (size_t)M * N + (size_t)i * M + r]
                        : G[p * (size_t)J * N + (size_t)i * J + (r - M)];
    }
    const size_t tr = (size_t)nprob * MJ;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tr; e += (size_t)gridDim.x * blockDim.x) {
        const size_t p = e / MJ;
        const int r = (int)(e - p * MJ);
        rhs[e] = (r < M) ? b[p * M + r] : g[p * J + (r - M)];
    }
}

void launch_prep(int nprob, int N, int M, int J, const double *A, const double *G, const double *b,
                 const double *g, double *Ct, double *rhs, hipStream_t stream) {
    if (M + J == 0 || nprob == 0) return;
    const size_t total = (size_t)nprob * (M + J) * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ssqp_prep_kernel, dim3(blocks), dim3(256), 0, stream, nprob, N, M, J, A, G, b, g, Ct, rhs);
}

hipError_t launch_solve(const SolveParams &P, int grid, size_t ldsBytes, hipStream_t stream) {
    hipError_t e;
    if ((P.N & 1) == 0) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ssqp_solve_kernel<2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(ssqp_solve_kernel<2>, dim3(grid), dim3(NT), ldsBytes, stream, P);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ssqp_solve_kernel<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(ssqp_solve_kernel<1>, dim3(grid), dim3(NT), ldsBytes, stream, P);
    }
    return hipGetLastError();
}

}  // namespace ssqp
