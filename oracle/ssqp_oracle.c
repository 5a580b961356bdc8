/*
 * ssqp_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the dense active-set
 * ("status switching") QP path of PharosAbad/StatusSwitchingQP.jl v1.0.2:
 *
 *     min 1/2 z'Vz + q'z   s.t.  Az = b,  Gz <= g,  d <= z <= u
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product (libssqp_hip.so) never links, loads or calls anything in oracle/.
 *
 * Parity pin: the reference is Julia and Julia is not installed in the build
 * image or on the GPU box, so the reference itself cannot be executed.  This
 * restatement is pinned by (i) the only known-answer test the reference holds
 * for this path, test/runtests.jl:23-32 (S == [UP, IN, IN]), (ii) an
 * independent numpy/scipy-LAPACK restatement (oracle/ssqp_numpy.py, which
 * calls the same LAPACK routines Julia's LinearAlgebra does: potrf/potri,
 * getrf/getri) and (iii) an independent KKT verifier (tests/kkt.py).  For any
 * input other than (i): "parity unpinned" by the reference's own tests.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/).  All matrices are column-major (Julia layout); indices
 * in the C code are 0-based, event ids in traces are 1-based like Julia's.
 *
 * The operation ORDER follows the reference (gathers by increasing index,
 * upper Cholesky A = U'U, explicit inverses inv(cholesky(.))), so that every
 * threshold decision is taken on a value computed the way the reference
 * computes it, up to the summation order inside BLAS/LAPACK kernels.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#include <stdio.h>
#endif

/* src/types.jl:17-23  @enum Status IN DN UP OE EO  (Int32 codes 0..4) */
enum { ST_IN = 0, ST_DN = 1, ST_UP = 2, ST_OE = 3, ST_EO = 4 };

/* src/types.jl:390-408  Settings{Float64} defaults */
typedef struct {
    int32_t maxIter; /* 7777 */
    int32_t rule;    /* 0 = :Dantzig (only rule restated) */
    double tol;      /* 2^-26 */
    double tolG;     /* 2^-33 */
} orc_settings;

/* detail codes written next to the solver status */
enum {
    ORC_OK = 0,
    ORC_POSDEF_V = 1,   /* cholesky(V[F,F]) would throw PosDefException, SSQP.jl:322 */
    ORC_POSDEF_C = 2,   /* cholesky(C) would throw PosDefException, SSQP.jl:328 */
    ORC_SINGULAR_LU = 3 /* lu() would throw SingularException, Simplex.jl:590 */
};

/* per-iteration trace record (optional) */
typedef struct {
    int32_t K;     /* free variables this iteration */
    int32_t W;     /* constraint rows kept after getRowsGJr */
    int32_t kind;  /* 0 K==0 pass, 1 blocked step, 2 release, 3 optimal */
    int32_t id;    /* first switched id (1-based; inequalities N+j), 0 if none */
} orc_trace;

#define IDX(i, j, ld) ((size_t)(i) + (size_t)(j) * (size_t)(ld))

/* ------------------------------------------------------------------------- */
/* optional LAPACK/BLAS arithmetic (cpu_baseline leg "lapack" of bench.py)     */
/* ------------------------------------------------------------------------- */
/* Julia's LinearAlgebra does the dense work of SSQP.jl:322-331,351-352 with OpenBLAS/LAPACK (potrf + potri,
 * gemm, gemv).  orc_lapack_load() binds the same routines from the OpenBLAS shared library that scipy bundles
 * (symbols scipy_dpotrf_ ...; plain dpotrf_ ... are tried too), so that the restatement can be TIMED with the
 * arithmetic the reference actually runs.  Decisions are the same threshold tests; only summation orders differ. */
#include <dlfcn.h>
typedef void (*orc_potrf_t)(const char *, const int *, double *, const int *, int *);
typedef void (*orc_gemm_t)(const char *, const char *, const int *, const int *, const int *, const double *,
                           const double *, const int *, const double *, const int *, const double *, double *,
                           const int *);
typedef void (*orc_gemv_t)(const char *, const int *, const int *, const double *, const double *, const int *,
                           const double *, const int *, const double *, double *, const int *);
static struct {
    void *h;
    orc_potrf_t potrf, potri;
    orc_gemm_t gemm;
    orc_gemv_t gemv;
} LP;
static __thread int t_lapack = 0; /* this thread's solves use LP */

static void *lp_sym(void *h, const char *name)
{
    char buf[64];
    snprintf(buf, sizeof buf, "scipy_%s", name);
    void *f = dlsym(h, buf);
    return f ? f : dlsym(h, name);
}
int orc_lapack_load(const char *path)
{
    if (LP.h) return 0;
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return 1;
    LP.potrf = (orc_potrf_t)lp_sym(h, "dpotrf_");
    LP.potri = (orc_potrf_t)lp_sym(h, "dpotri_");
    LP.gemm = (orc_gemm_t)lp_sym(h, "dgemm_");
    LP.gemv = (orc_gemv_t)lp_sym(h, "dgemv_");
    void (*setn)(int) = (void (*)(int))lp_sym(h, "openblas_set_num_threads");
    if (!LP.potrf || !LP.potri || !LP.gemm || !LP.gemv) return 2;
    if (setn) setn(1); /* one BLAS thread per QP: the parallelism is one QP per core */
    LP.h = h;
    return 0;
}
static void lp_gemm(char ta, char tb, int m, int n, int k, double al, const double *a, int lda, const double *b,
                    int ldb, double be, double *c, int ldc)
{
    if (m <= 0 || n <= 0) return;
    if (k <= 0) {
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < m; ++i) c[IDX(i, j, ldc)] = (be == 0.0) ? 0.0 : be * c[IDX(i, j, ldc)];
        return;
    }
    LP.gemm(&ta, &tb, &m, &n, &k, &al, a, &lda, b, &ldb, &be, c, &ldc);
}
static void lp_gemv(char t, int m, int n, double al, const double *a, int lda, const double *x, double be, double *y)
{
    const int one = 1;
    const int ylen = (t == 'N') ? m : n;
    if (ylen <= 0) return;
    if (m <= 0 || n <= 0) {
        for (int i = 0; i < ylen; ++i) y[i] = (be == 0.0) ? 0.0 : be * y[i];
        return;
    }
    LP.gemv(&t, &m, &n, &al, a, &lda, x, &one, &be, y, &one);
}
/* inv(cholesky(A)) by potrf('U') + potri('U') + mirror; returns potrf's info */
static int lp_chol_inverse(double *a, int n)
{
    int info = 0;
    const char U = 'U';
    if (n <= 0) return 0;
    LP.potrf(&U, &n, a, &n, &info);
    if (info != 0) return info;
    LP.potri(&U, &n, a, &n, &info);
    for (int j = 0; j < n; ++j)
        for (int i = j + 1; i < n; ++i) a[IDX(i, j, n)] = a[IDX(j, i, n)];
    return info;
}

/* ------------------------------------------------------------------------- */
/* dense helpers (what Julia delegates to LinearAlgebra / LAPACK)             */
/* ------------------------------------------------------------------------- */

/* cholesky(A) for a dense symmetric matrix: LAPACK potrf('U'), A = U'U.
 * Overwrites the upper triangle of a (n x n, ld n).  Returns 0 or the 1-based
 * order of the failing leading minor (Julia throws PosDefException(info)). */
static int chol_upper(double *a, int n)
{
    for (int j = 0; j < n; ++j) {
        double s = a[IDX(j, j, n)];
        for (int k = 0; k < j; ++k) s -= a[IDX(k, j, n)] * a[IDX(k, j, n)];
        if (!(s > 0.0)) return j + 1;
        double ujj = sqrt(s);
        a[IDX(j, j, n)] = ujj;
        for (int i = j + 1; i < n; ++i) {
            double t = a[IDX(j, i, n)];
            for (int k = 0; k < j; ++k) t -= a[IDX(k, j, n)] * a[IDX(k, i, n)];
            a[IDX(j, i, n)] = t / ujj;
        }
    }
    return 0;
}

/* inv(cholesky(A)): LAPACK potri('U') = trtri (inv U) then lauum
 * (inv(U) * inv(U)'), then the triangle is mirrored (LinearAlgebra copytri!).
 * On entry the upper triangle of a holds U; on exit a = inv(A), full. */
static void chol_upper_inverse(double *a, int n)
{
    /* trtri: U := inv(U), column by column */
    for (int j = 0; j < n; ++j) {
        a[IDX(j, j, n)] = 1.0 / a[IDX(j, j, n)];
        double ajj = -a[IDX(j, j, n)];
        /* x = U(0:j-1,0:j-1)^{-1-already-inverted} * U(0:j-1, j) */
        for (int i = 0; i < j; ++i) {
            double s = 0.0;
            for (int k = i; k < j; ++k) s += a[IDX(i, k, n)] * a[IDX(k, j, n)];
            a[IDX(i, j, n)] = s; /* uses old a[k][j], k>i: safe, ascending i */
        }
        for (int i = 0; i < j; ++i) a[IDX(i, j, n)] *= ajj;
    }
    /* lauum: upper triangle of Uinv * Uinv' */
    for (int i = 0; i < n; ++i) {
        for (int j = i; j < n; ++j) {
            double s = 0.0;
            for (int k = j; k < n; ++k) s += a[IDX(i, k, n)] * a[IDX(j, k, n)];
            a[IDX(i, j, n)] = s; /* row i only reads columns >= j of rows i, j>=i */
        }
    }
    for (int j = 0; j < n; ++j)
        for (int i = j + 1; i < n; ++i) a[IDX(i, j, n)] = a[IDX(j, i, n)];
}

/* inv(lu(A)): getrf (partial pivoting) + getri.  a (n x n) is overwritten by
 * inv(A).  Returns 0, or k>0 when U[k,k]==0 (Julia: SingularException). */
static int lu_inverse(double *a, int n, double *work, int *piv)
{
    for (int k = 0; k < n; ++k) {
        int p = k;
        double m = fabs(a[IDX(k, k, n)]);
        for (int i = k + 1; i < n; ++i) {
            double v = fabs(a[IDX(i, k, n)]);
            if (v > m) { m = v; p = i; }
        }
        piv[k] = p;
        if (m == 0.0) return k + 1;
        if (p != k)
            for (int j = 0; j < n; ++j) {
                double t = a[IDX(k, j, n)];
                a[IDX(k, j, n)] = a[IDX(p, j, n)];
                a[IDX(p, j, n)] = t;
            }
        double inv = 1.0 / a[IDX(k, k, n)];
        for (int i = k + 1; i < n; ++i) a[IDX(i, k, n)] *= inv;
        for (int j = k + 1; j < n; ++j) {
            double t = a[IDX(k, j, n)];
            for (int i = k + 1; i < n; ++i) a[IDX(i, j, n)] -= a[IDX(i, k, n)] * t;
        }
    }
    /* X = inv(P'LU) : solve L U X = P I column by column into work */
    for (int c = 0; c < n; ++c) {
        double *x = work + (size_t)c * n;
        for (int i = 0; i < n; ++i) x[i] = 0.0;
        x[c] = 1.0;
        for (int k = 0; k < n; ++k)
            if (piv[k] != k) { double t = x[k]; x[k] = x[piv[k]]; x[piv[k]] = t; }
        for (int k = 0; k < n; ++k) {
            double t = x[k];
            if (t != 0.0)
                for (int i = k + 1; i < n; ++i) x[i] -= a[IDX(i, k, n)] * t;
        }
        for (int k = n - 1; k >= 0; --k) {
            x[k] /= a[IDX(k, k, n)];
            double t = x[k];
            for (int i = 0; i < k; ++i) x[i] -= a[IDX(i, k, n)] * t;
        }
    }
    memcpy(a, work, sizeof(double) * (size_t)n * n);
    return 0;
}

/* x = A \ y for a tall full-column-rank A (m x n, m >= n): least squares by
 * Householder QR (Julia's `\` on a rectangular matrix uses pivoted QR; for a
 * full-column-rank A the solution is unique, SSQP.jl:158).  a, y overwritten. */
static void lstsq_qr(double *a, int m, int n, double *y, double *x)
{
    for (int k = 0; k < n; ++k) {
        double nrm = 0.0;
        for (int i = k; i < m; ++i) nrm += a[IDX(i, k, m)] * a[IDX(i, k, m)];
        nrm = sqrt(nrm);
        if (nrm == 0.0) continue;
        double alpha = a[IDX(k, k, m)] > 0 ? -nrm : nrm;
        double v0 = a[IDX(k, k, m)] - alpha;
        a[IDX(k, k, m)] = alpha;
        /* v = [v0; a[k+1:m,k]], beta = 2/(v'v) */
        double vtv = v0 * v0;
        for (int i = k + 1; i < m; ++i) vtv += a[IDX(i, k, m)] * a[IDX(i, k, m)];
        if (vtv == 0.0) continue;
        double beta = 2.0 / vtv;
        for (int j = k + 1; j < n; ++j) {
            double s = v0 * a[IDX(k, j, m)];
            for (int i = k + 1; i < m; ++i) s += a[IDX(i, k, m)] * a[IDX(i, j, m)];
            s *= beta;
            a[IDX(k, j, m)] -= s * v0;
            for (int i = k + 1; i < m; ++i) a[IDX(i, j, m)] -= s * a[IDX(i, k, m)];
        }
        double s = v0 * y[k];
        for (int i = k + 1; i < m; ++i) s += a[IDX(i, k, m)] * y[i];
        s *= beta;
        y[k] -= s * v0;
        for (int i = k + 1; i < m; ++i) y[i] -= s * a[IDX(i, k, m)];
    }
    for (int k = n - 1; k >= 0; --k) {
        double s = y[k];
        for (int j = k + 1; j < n; ++j) s -= a[IDX(k, j, m)] * x[j];
        x[k] = s / a[IDX(k, k, m)];
    }
}

/* ------------------------------------------------------------------------- */
/* src/utils.jl:49-86  getRowsGJr(X, tol): row-wise Gauss-Jordan rank filter  */
/* ------------------------------------------------------------------------- */
/* X is nr x nc column-major (not modified).  rows[] receives the 0-based kept
 * rows (increasing); returns their count.  *l1 as the reference (= count). */
int orc_getRowsGJr(const double *X, int nr, int nc, double tol, int *rows, int *l1out)
{
    double *A = (double *)malloc(sizeof(double) * (size_t)nr * nc + 8);
    int *c0 = (int *)malloc(sizeof(int) * (size_t)(nc + 1));
    memcpy(A, X, sizeof(double) * (size_t)nr * nc);
    for (int c = 0; c < nc; ++c) c0[c] = c;
    int nrows = 0, l1 = 0;
    int i = 0, j = 0;
    while (i < nr && j < nc) { /* utils.jl:58 */
        /* findmax(abs.(A[i, c0[j:nc]])): first maximum in c0 order, :59 */
        int mj = j;
        double m = fabs(A[IDX(i, c0[j], nr)]);
        for (int t = j + 1; t < nc; ++t) {
            double v = fabs(A[IDX(i, c0[t], nr)]);
            if (v > m) { m = v; mj = t; }
        }
        if (m <= tol) { /* :61 */
            i += 1;
        } else {
            rows[nrows++] = i;                          /* :64 */
            int t = c0[mj]; c0[mj] = c0[j]; c0[j] = t;  /* :65 */
            int n = c0[j];
            double dd = A[IDX(i, n, nr)];
            for (int t2 = j; t2 < nc; ++t2) A[IDX(i, c0[t2], nr)] /= dd; /* :68-70 */
            for (int k = 0; k < nr; ++k) {                               /* :71-78 */
                if (k == i) continue;
                double dk = A[IDX(k, n, nr)];
                for (int t2 = j; t2 < nc; ++t2)
                    A[IDX(k, c0[t2], nr)] -= dk * A[IDX(i, c0[t2], nr)];
            }
            l1 = j + 1; /* :79 (1-based j) */
            i += 1;
            j += 1;
        }
    }
    free(A);
    free(c0);
    if (l1out) *l1out = l1;
    return nrows;
}

/* ------------------------------------------------------------------------- */
/* src/SSQP.jl:10-32  polishSz!                                               */
/* ------------------------------------------------------------------------- */
static void polishSz(int32_t *S, double *z, const double *d, const double *u, const double *G,
                     const double *g, int N, int J, double tol)
{
    for (int k = 0; k < N; ++k) {
        if (S[k] == ST_DN) z[k] = d[k];
        else if (S[k] == ST_UP) z[k] = u[k];
        else {
            if (fabs(z[k] - d[k]) < tol) { z[k] = d[k]; S[k] = ST_DN; }
            else if (fabs(z[k] - u[k]) < tol) { z[k] = u[k]; S[k] = ST_UP; }
        }
    }
    for (int j = 0; j < J; ++j) { /* :28-30 */
        double s = 0.0;
        for (int k = 0; k < N; ++k) s += z[k] * G[IDX(j, k, J)];
        S[N + j] = fabs(g[j] - s) < tol ? ST_EO : ST_OE;
    }
}

/* ------------------------------------------------------------------------- */
/* src/SSQP.jl:35-59  freeK!  (K == 0 branch)                                 */
/* ------------------------------------------------------------------------- */
static int freeK(int32_t *S, const double *z, const double *V, const double *q, int N, double tol,
                 double *p, int32_t *S0)
{
    for (int i = 0; i < N; ++i) p[i] = 0.0;
    for (int j = 0; j < N; ++j) { /* p = V*z + q  (gemv, column sweep) */
        double zj = z[j];
        if (zj != 0.0)
            for (int i = 0; i < N; ++i) p[i] += V[IDX(i, j, N)] * zj;
    }
    for (int i = 0; i < N; ++i) p[i] += q[i];
    memcpy(S0, S, sizeof(int32_t) * (size_t)N);
    int t = 1;
    for (int k = 0; k < N; ++k) { /* :41-47 */
        if ((p[k] >= -tol && S[k] == ST_UP) || (p[k] <= tol && S[k] == ST_DN)) {
            S[k] = ST_IN;
            t = 0;
        }
    }
    if (t) return 1;
    int nip = 0;
    double nrm = 0.0;
    for (int k = 0; k < N; ++k)
        if (S[k] == ST_IN) { nip++; if (fabs(p[k]) > nrm) nrm = fabs(p[k]); }
    if (nip > 0 && nrm <= tol) { /* :52-55 */
        for (int k = 0; k < N; ++k)
            if (S[k] == ST_IN) S[k] = S0[k];
        return 1;
    }
    return -1;
}

typedef struct { int32_t from, to, id; double L; } event_t; /* types.jl:39-44 */

/* stable "sort!(Lo, by = x -> x.L)" is only used for its first element and a
 * threshold pass (SSQP.jl:94-116, 176-177): first minimum in push order. */
static int first_min(const event_t *ev, int n)
{
    int im = 0;
    for (int i = 1; i < n; ++i)
        if (ev[i].L < ev[im].L) im = i;
    return im;
}

/* ------------------------------------------------------------------------- */
/* the solver                                                                 */
/* ------------------------------------------------------------------------- */

typedef struct {
    int N, M, J;
    /* index lists */
    int *iF, *iB, *iEg, *iOg, *ra;
    /* dense work */
    double *AE, *AB, *bE, *X, *zB, *VFF, *c, *mT, *C, *TC, *VQ, *alpha, *p, *alphaL, *gamma;
    double *tmpW, *tmpK, *lsA, *lsy, *lsx, *pN;
    double *VBF, *VBB; /* LAPACK mode: the gathers V[B,F], V[B,B] the reference materialises (SSQP.jl:323,352) */
    int32_t *S0;
    event_t *ev;
} work_t;

static void *xm(size_t n) { void *p = malloc(n ? n : 8); if (!p) abort(); return p; }

static void work_alloc(work_t *w, int N, int M, int J)
{
    int W0 = M + J;
    w->N = N; w->M = M; w->J = J;
    w->iF = xm(sizeof(int) * N); w->iB = xm(sizeof(int) * N);
    w->iEg = xm(sizeof(int) * (J + 1)); w->iOg = xm(sizeof(int) * (J + 1));
    w->ra = xm(sizeof(int) * (W0 + 1));
    w->AE = xm(sizeof(double) * (size_t)W0 * N); w->AB = xm(sizeof(double) * (size_t)W0 * N);
    w->bE = xm(sizeof(double) * (W0 + 1));
    w->X = xm(sizeof(double) * (size_t)W0 * (N + 1));
    w->zB = xm(sizeof(double) * N);
    w->VFF = xm(sizeof(double) * (size_t)N * N);
    w->c = xm(sizeof(double) * N);
    w->mT = xm(sizeof(double) * (size_t)N * (W0 + 1));
    w->C = xm(sizeof(double) * (size_t)(W0 + 1) * (W0 + 1));
    w->TC = xm(sizeof(double) * (size_t)N * (W0 + 1));
    w->VQ = xm(sizeof(double) * (size_t)N * N);
    w->alpha = xm(sizeof(double) * N); w->p = xm(sizeof(double) * N);
    w->alphaL = xm(sizeof(double) * (W0 + 1)); w->gamma = xm(sizeof(double) * N);
    w->tmpW = xm(sizeof(double) * (W0 + 1)); w->tmpK = xm(sizeof(double) * N);
    w->lsA = xm(sizeof(double) * (size_t)N * (W0 + 1));
    w->lsy = xm(sizeof(double) * N); w->lsx = xm(sizeof(double) * (W0 + 1));
    w->pN = xm(sizeof(double) * N);
    w->VBF = t_lapack ? xm(sizeof(double) * (size_t)N * N) : NULL;
    w->VBB = t_lapack ? xm(sizeof(double) * (size_t)N * N) : NULL;
    w->S0 = xm(sizeof(int32_t) * (N + J));
    w->ev = xm(sizeof(event_t) * (size_t)(N + J + 1));
}

static void work_free(work_t *w)
{
    free(w->iF); free(w->iB); free(w->iEg); free(w->iOg); free(w->ra);
    free(w->AE); free(w->AB); free(w->bE); free(w->X); free(w->zB); free(w->VFF);
    free(w->c); free(w->mT); free(w->C); free(w->TC); free(w->VQ); free(w->alpha);
    free(w->p); free(w->alphaL); free(w->gamma); free(w->tmpW); free(w->tmpK);
    free(w->lsA); free(w->lsy); free(w->lsx); free(w->pN); free(w->S0); free(w->ev);
    free(w->VBF); free(w->VBB);
}

/* src/SSQP.jl:61-134  aStep!  -- returns -1 (blocked) or +1 (full step) */
static int aStep(work_t *w, const double *p, double *z, int32_t *S, int K, int nOg,
                 const double *alpha, const double *G, const double *g, const double *d,
                 const double *u, int N, int J, double tol, int *first_id)
{
    event_t *Lo = w->ev;
    int nL = 0;
    for (int k = 0; k < K; ++k) { /* :65-76 */
        int j = w->iF[k];
        double t = p[k], h = z[j];
        double dL = (d[j] - h) / t, uL = (u[j] - h) / t;
        if (t > tol && u[j] < INFINITY) Lo[nL++] = (event_t){ST_IN, ST_UP, j + 1, uL};
        else if (t < -tol && d[j] > -INFINITY) Lo[nL++] = (event_t){ST_IN, ST_DN, j + 1, dL};
    }
    if (J > 0) { /* :78-89 */
        for (int k = 0; k < nOg; ++k) {
            int j = w->iOg[k];
            double zo = 0.0, po = 0.0;
            for (int i = 0; i < N; ++i) zo += G[IDX(j, i, J)] * z[i];
            zo = g[j] - zo;
            for (int i = 0; i < K; ++i) po += G[IDX(j, w->iF[i], J)] * p[i];
            if (po > tol) Lo[nL++] = (event_t){ST_OE, ST_EO, j + 1, zo / po};
        }
    }
    double L1 = 1.0; /* :91-96 */
    if (nL > 0) L1 = Lo[first_min(Lo, nL)].L;
    if (L1 < 1.0) { /* :98-127 */
        for (int k = 0; k < K; ++k) z[w->iF[k]] += L1 * p[k];
        *first_id = 0;
        for (int i = 0; i < nL; ++i) {
            if (Lo[i].L - L1 > tol) continue; /* sorted order + break == this filter */
            int k = Lo[i].id;                 /* 1-based */
            int To = Lo[i].to;
            if (To == ST_EO) k += N;
            S[k - 1] = To;
            if (k <= N) z[k - 1] = (To == ST_DN) ? d[k - 1] : u[k - 1];
            if (*first_id == 0 || k < *first_id) *first_id = k;
        }
        return -1;
    }
    for (int k = 0; k < K; ++k) z[w->iF[k]] = alpha[k]; /* :130 */
    return 1;
}

/*
 * src/SSQP.jl:237-377  solveQP(Q, S, x0; settings)
 *
 * S (N+J, in/out) and z (N, out; x0 is copied into it, :266).  Returns the
 * reference's `status`: iter > 0 on success, -iter when iter > maxIter (:273),
 * -1 for a numerical error (:314; also where Julia would throw, see *detail).
 * trace (may be NULL) receives up to ntrace records; *ntrace_out the count.
 */
/* The multipliers of the LAST pass (may be NULL), laid out by row / variable id so that a caller can compare them
 * without knowing the working set:
 *   lambda (M+J): alphaL (:351) of every kept row; for an active inequality that the rank filter purged, the value
 *                 KKTchk! computes for it (:158-159); 0 for inactive inequalities and purged equality rows (the
 *                 reference computes nothing for those)
 *   gamma (N):    gamma (:352) of the bound variables, 0 for the free ones
 * written only on the successful exit (status > 0); on the K == 0 exit (:278-285, no multipliers exist in the
 * reference) gamma = V z + q (what freeK! tests) and lambda = 0. */
int64_t orc_solveQP_warm_ex(int N, int M, int J, const double *V, const double *A, const double *G,
                            const double *q, const double *b, const double *g, const double *d,
                            const double *u, int32_t *S, const double *x0, double *z,
                            const orc_settings *st, int32_t *detail, orc_trace *trace, int ntrace,
                            int *ntrace_out, double *lambda_out, double *gamma_out);

int64_t orc_solveQP_warm(int N, int M, int J, const double *V, const double *A, const double *G,
                         const double *q, const double *b, const double *g, const double *d,
                         const double *u, int32_t *S, const double *x0, double *z,
                         const orc_settings *st, int32_t *detail, orc_trace *trace, int ntrace,
                         int *ntrace_out)
{
    return orc_solveQP_warm_ex(N, M, J, V, A, G, q, b, g, d, u, S, x0, z, st, detail, trace, ntrace, ntrace_out,
                               NULL, NULL);
}

int64_t orc_solveQP_warm_ex(int N, int M, int J, const double *V, const double *A, const double *G,
                            const double *q, const double *b, const double *g, const double *d,
                            const double *u, int32_t *S, const double *x0, double *z,
                            const orc_settings *st, int32_t *detail, orc_trace *trace, int ntrace,
                            int *ntrace_out, double *lambda_out, double *gamma_out)
{
    const int maxIter = st->maxIter;
    const double tol = st->tol, tolG = st->tolG;
    work_t wk, *w = &wk;
    work_alloc(w, N, M, J);
    if (detail) *detail = ORC_OK;
    int nt = 0;
    int64_t ret = 0;

    memcpy(z, x0, sizeof(double) * (size_t)N); /* :266 */
    int64_t iter = 0;
    for (;;) {
        iter += 1; /* :271-274 */
        if (iter > maxIter) { ret = -iter; break; }

        int K = 0, R = 0;
        for (int j = 0; j < N; ++j) { /* :276-277, :287 */
            if (S[j] == ST_IN) w->iF[K++] = j; else w->iB[R++] = j;
        }
        if (K == 0) { /* :278-285 */
            int s = freeK(S, z, V, q, N, tol, w->pN, w->S0);
            if (trace && nt < ntrace) trace[nt] = (orc_trace){0, 0, s > 0 ? 3 : 0, 0};
            nt++;
            if (s > 0) {
                if (lambda_out) for (int r = 0; r < M + J; ++r) lambda_out[r] = 0.0;
                if (gamma_out) for (int j = 0; j < N; ++j) gamma_out[j] = w->pN[j];
                ret = iter;
                break;
            }
            continue;
        }
        int JE = 0, nOg = 0;
        for (int j = 0; j < J; ++j) { /* :288-289 */
            if (S[N + j] == ST_EO) w->iEg[JE++] = j;
            else if (S[N + j] == ST_OE) w->iOg[nOg++] = j;
        }
        int W0 = M + JE;
        /* AE = [A[:,F]; G[Eg,F]], AB = [A[:,B]; G[Eg,B]]   :290-294 */
        double *AE = w->AE, *AB = w->AB, *bE = w->bE;
        for (int k = 0; k < K; ++k) {
            int j = w->iF[k];
            for (int r = 0; r < M; ++r) AE[IDX(r, k, W0)] = A[IDX(r, j, M)];
            for (int r = 0; r < JE; ++r) AE[IDX(M + r, k, W0)] = G[IDX(w->iEg[r], j, J)];
        }
        for (int k = 0; k < R; ++k) {
            int j = w->iB[k];
            for (int r = 0; r < M; ++r) AB[IDX(r, k, W0)] = A[IDX(r, j, M)];
            for (int r = 0; r < JE; ++r) AB[IDX(M + r, k, W0)] = G[IDX(w->iEg[r], j, J)];
            w->zB[k] = z[j];
        }
        /* bE = [b; g[Eg]] - AB*zB   :295 */
        for (int r = 0; r < M; ++r) bE[r] = 0.0;
        for (int r = 0; r < W0; ++r) w->tmpW[r] = 0.0;
        for (int k = 0; k < R; ++k) {
            double zk = w->zB[k];
            if (zk != 0.0)
                for (int r = 0; r < W0; ++r) w->tmpW[r] += AB[IDX(r, k, W0)] * zk;
        }
        for (int r = 0; r < M; ++r) bE[r] = b[r] - w->tmpW[r];
        for (int r = 0; r < JE; ++r) bE[M + r] = g[w->iEg[r]] - w->tmpW[M + r];

        /* ra, la = getRowsGJr([AE bE], tol)   :310-319 */
        int W = W0;
        {
            double *X = w->X;
            memcpy(X, AE, sizeof(double) * (size_t)W0 * K);
            for (int r = 0; r < W0; ++r) X[IDX(r, K, W0)] = bE[r];
            int la = 0;
            W = orc_getRowsGJr(X, W0, K + 1, tol, w->ra, &la);
            if (W < W0) {
                /* W != la is dead code (la == W always), :313-315 */
                /* compact rows in place (ra increasing) */
                double *AE2 = w->X; /* reuse X as temp */
                for (int k = 0; k < K; ++k)
                    for (int r = 0; r < W; ++r) AE2[IDX(r, k, W)] = AE[IDX(w->ra[r], k, W0)];
                memcpy(AE, AE2, sizeof(double) * (size_t)W * K);
                for (int k = 0; k < R; ++k)
                    for (int r = 0; r < W; ++r) AE2[IDX(r, k, W)] = AB[IDX(w->ra[r], k, W0)];
                memcpy(AB, AE2, sizeof(double) * (size_t)W * R);
                for (int r = 0; r < W; ++r) w->tmpW[r] = bE[w->ra[r]];
                for (int r = 0; r < W; ++r) bE[r] = w->tmpW[r];
            }
        }

        /* iV = inv(cholesky(V[F,F]))   :322 */
        double *iV = w->VFF;
        for (int b2 = 0; b2 < K; ++b2)
            for (int a2 = 0; a2 < K; ++a2) iV[IDX(a2, b2, K)] = V[IDX(w->iF[a2], w->iF[b2], N)];
        double *mT = w->mT, *C = w->C, *TC = w->TC, *VQ = w->VQ;
        double pinf = 0.0;
        int pnan = 0;
        if (t_lapack) { /* the same statements with LAPACK/BLAS doing the arithmetic, as LinearAlgebra does */
            if (lp_chol_inverse(iV, K) != 0) {
                if (detail) *detail = ORC_POSDEF_V;
                ret = -1;
                break;
            }
            /* VBF = V[B,F]; c = VBF'*zB + q[F]   :323-324 */
            for (int k = 0; k < K; ++k) {
                const double *col = V + (size_t)w->iF[k] * N;
                for (int r = 0; r < R; ++r) w->VBF[IDX(r, k, R)] = col[w->iB[r]];
                w->c[k] = q[w->iF[k]];
            }
            lp_gemv('T', R, K, 1.0, w->VBF, R > 0 ? R : 1, w->zB, 1.0, w->c);
            lp_gemm('N', 'T', K, W, K, 1.0, iV, K, AE, W > 0 ? W : 1, 0.0, mT, K);   /* mT = iV*AE'   :325 */
            lp_gemm('N', 'N', W, W, K, 1.0, AE, W > 0 ? W : 1, mT, K, 0.0, C, W > 0 ? W : 1); /* C = AE*mT :326 */
            for (int c2 = 0; c2 < W; ++c2)
                for (int r = c2; r < W; ++r) {
                    double sv = (C[IDX(r, c2, W)] + C[IDX(c2, r, W)]) / 2;
                    C[IDX(r, c2, W)] = sv;
                    C[IDX(c2, r, W)] = sv;
                }
            if (W > 0 && lp_chol_inverse(C, W) != 0) {
                if (detail) *detail = ORC_POSDEF_C;
                ret = -1;
                break;
            }
            lp_gemm('N', 'N', K, W, W, 1.0, mT, K, C, W > 0 ? W : 1, 0.0, TC, K);   /* TC = mT*C   :329 */
            memcpy(VQ, iV, sizeof(double) * (size_t)K * K);
            lp_gemm('N', 'T', K, K, W, -1.0, mT, K, TC, K, 1.0, VQ, K);             /* VQ = iV - mT*TC'   :330 */
            lp_gemv('N', K, W, 1.0, TC, K, bE, 0.0, w->alpha);                      /* alpha = TC*bE - VQ*c :331 */
            lp_gemv('N', K, K, -1.0, VQ, K, w->c, 1.0, w->alpha);
            for (int i = 0; i < K; ++i) {
                w->p[i] = w->alpha[i] - z[w->iF[i]];
                double a = fabs(w->p[i]);
                if (a != a) pnan = 1;
                if (a > pinf) pinf = a;
            }
        } else {
        if (chol_upper(iV, K) != 0) {
            if (detail) *detail = ORC_POSDEF_V;
            ret = -1;
            break;
        }
        chol_upper_inverse(iV, K);
        /* c = V[B,F]'*zB + q[F]   :323-324 */
        for (int k = 0; k < K; ++k) {
            const double *col = V + (size_t)w->iF[k] * N; /* V[B,F[k]] = V[:,F[k]][B] */
            double s = 0.0;
            for (int r = 0; r < R; ++r) s += col[w->iB[r]] * w->zB[r];
            w->c[k] = s + q[w->iF[k]];
        }
        /* mT = iV*AE' (K x W)   :325 */
        for (int r = 0; r < W; ++r)
            for (int i = 0; i < K; ++i) {
                double s = 0.0;
                for (int k = 0; k < K; ++k) s += iV[IDX(i, k, K)] * AE[IDX(r, k, W)];
                mT[IDX(i, r, K)] = s;
            }
        /* C = AE*mT; C = (C+C')/2; C = inv(cholesky(C))   :326-328 */
        for (int c2 = 0; c2 < W; ++c2)
            for (int r = 0; r < W; ++r) {
                double s = 0.0;
                for (int k = 0; k < K; ++k) s += AE[IDX(r, k, W)] * mT[IDX(k, c2, K)];
                C[IDX(r, c2, W)] = s;
            }
        for (int c2 = 0; c2 < W; ++c2)
            for (int r = c2; r < W; ++r) {
                double s = (C[IDX(r, c2, W)] + C[IDX(c2, r, W)]) / 2;
                C[IDX(r, c2, W)] = s;
                C[IDX(c2, r, W)] = s;
            }
        if (W > 0) {
            if (chol_upper(C, W) != 0) {
                if (detail) *detail = ORC_POSDEF_C;
                ret = -1;
                break;
            }
            chol_upper_inverse(C, W);
        }
        /* TC = mT*C (K x W)   :329 */
        for (int c2 = 0; c2 < W; ++c2)
            for (int i = 0; i < K; ++i) {
                double s = 0.0;
                for (int r = 0; r < W; ++r) s += mT[IDX(i, r, K)] * C[IDX(r, c2, W)];
                TC[IDX(i, c2, K)] = s;
            }
        /* VQ = iV - mT*TC'   :330 */
        for (int j2 = 0; j2 < K; ++j2)
            for (int i = 0; i < K; ++i) {
                double s = 0.0;
                for (int r = 0; r < W; ++r) s += mT[IDX(i, r, K)] * TC[IDX(j2, r, K)];
                VQ[IDX(i, j2, K)] = iV[IDX(i, j2, K)] - s;
            }
        /* alpha = TC*bE - VQ*c ; p = alpha - z[F]   :331-332 */
        for (int i = 0; i < K; ++i) {
            double s1 = 0.0, s2 = 0.0;
            for (int r = 0; r < W; ++r) s1 += TC[IDX(i, r, K)] * bE[r];
            for (int k = 0; k < K; ++k) s2 += VQ[IDX(i, k, K)] * w->c[k];
            w->alpha[i] = s1 - s2;
            w->p[i] = w->alpha[i] - z[w->iF[i]];
            double a = fabs(w->p[i]);
            if (a != a) pnan = 1;
            if (a > pinf) pinf = a;
        }
        } /* !t_lapack */
        if (pnan) pinf = NAN; /* norm(p, Inf) propagates NaN */

        if (pinf > tolG) { /* :335-340 */
            int fid = 0;
            int s = aStep(w, w->p, z, S, K, nOg, w->alpha, G, g, d, u, N, J, tol, &fid);
            if (s < 0) {
                if (trace && nt < ntrace) trace[nt] = (orc_trace){K, W, 1, fid};
                nt++;
                continue;
            }
        }
        if (t_lapack) {
            /* alphaL = -(TC'*c + C*bE)   :351 */
            lp_gemv('T', K, W, -1.0, TC, K, w->c, 0.0, w->alphaL);
            lp_gemv('N', W, W, -1.0, C, W > 0 ? W : 1, bE, 1.0, w->alphaL);
            /* gamma = VBF*alpha + V[B,B]*zB + q[B] + AB'*alphaL   :352 (V[B,B] is gathered like the reference does) */
            for (int k = 0; k < R; ++k) {
                const double *col = V + (size_t)w->iB[k] * N;
                for (int r = 0; r < R; ++r) w->VBB[IDX(r, k, R)] = col[w->iB[r]];
                w->gamma[k] = q[w->iB[k]];
            }
            lp_gemv('N', R, K, 1.0, w->VBF, R > 0 ? R : 1, w->alpha, 1.0, w->gamma);
            lp_gemv('N', R, R, 1.0, w->VBB, R > 0 ? R : 1, w->zB, 1.0, w->gamma);
            lp_gemv('T', W, R, 1.0, AB, W > 0 ? W : 1, w->alphaL, 1.0, w->gamma);
        } else {
        /* alphaL = -(TC'*c + C*bE)   :351 */
        for (int r = 0; r < W; ++r) {
            double s = 0.0;
            for (int k = 0; k < K; ++k) s += TC[IDX(k, r, K)] * w->c[k];
            double s2 = 0.0;
            for (int r2 = 0; r2 < W; ++r2) s2 += C[IDX(r, r2, W)] * bE[r2];
            w->alphaL[r] = -(s + s2);
        }
        /* gamma = VBF*alpha + V[B,B]*zB + q[B] + AB'*alphaL   :352 */
        for (int r = 0; r < R; ++r) {
            int i = w->iB[r];
            const double *col = V + (size_t)i * N; /* row i of V == column i (symmetric) */
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            for (int k = 0; k < K; ++k) s1 += col[w->iF[k]] * w->alpha[k];
            for (int k = 0; k < R; ++k) s2 += col[w->iB[k]] * w->zB[k];
            for (int r2 = 0; r2 < W; ++r2) s3 += AB[IDX(r2, r, W)] * w->alphaL[r2];
            w->gamma[r] = ((s1 + s2) + q[i]) + s3;
        }
        } /* !t_lapack */
        /* KKTchk!   :136-188 */
        event_t *Li = w->ev;
        int nL = 0;
        for (int k = 0; k < R; ++k) { /* :139-147 */
            int j = w->iB[k];
            double t = w->gamma[k];
            if (S[j] == ST_UP && t > tolG) Li[nL++] = (event_t){ST_UP, ST_IN, j + 1, -t};
            else if (S[j] == ST_DN && t < -tolG) Li[nL++] = (event_t){ST_DN, ST_IN, j + 1, t};
        }
        if (lambda_out) {
            for (int r = 0; r < M + J; ++r) lambda_out[r] = 0.0;
            for (int r = 0; r < W; ++r) { /* kept rows: position r is row ra[r] of [A; G[Eg,:]] */
                int row = (W == W0) ? r : w->ra[r];
                lambda_out[row < M ? row : M + w->iEg[row - M]] = w->alphaL[r];
            }
        }
        if (gamma_out) {
            for (int j = 0; j < N; ++j) gamma_out[j] = 0.0;
            for (int k = 0; k < R; ++k) gamma_out[w->iB[k]] = w->gamma[k];
        }
        if (JE > 0) { /* :150-172 */
            for (int j = 0; j < JE; ++j) {
                /* position of active inequality j among the kept rows */
                int pos = -1;
                for (int r = 0; r < W; ++r)
                    if (w->ra[r] == M + j) { pos = r; break; }
                if (W == W0) pos = M + j; /* ra untouched when nothing was purged */
                double Lda;
                if (pos < 0) { /* purged row: x = AE' \ GE[j,F]; Lda = alphaL'x   :158-159 */
                    for (int k = 0; k < K; ++k) {
                        for (int r = 0; r < W; ++r) w->lsA[IDX(k, r, K)] = AE[IDX(r, k, W)];
                        w->lsy[k] = G[IDX(w->iEg[j], w->iF[k], J)];
                    }
                    for (int r = 0; r < W; ++r) w->lsx[r] = 0.0;
                    if (K >= W) lstsq_qr(w->lsA, K, W, w->lsy, w->lsx);
                    Lda = 0.0;
                    for (int r = 0; r < W; ++r) Lda += w->alphaL[r] * w->lsx[r];
                } else {
                    Lda = w->alphaL[pos];
                }
                if (lambda_out) lambda_out[M + w->iEg[j]] = Lda;
                if (Lda < -tolG) Li[nL++] = (event_t){ST_EO, ST_OE, w->iEg[j] + 1, Lda};
            }
        }
        if (nL > 0) { /* :175-184 */
            event_t e = Li[first_min(Li, nL)];
            int k = e.id;
            if (e.to == ST_OE) k += N;
            S[k - 1] = e.to;
            if (trace && nt < ntrace) trace[nt] = (orc_trace){K, W, 2, k};
            nt++;
            continue;
        }
        if (trace && nt < ntrace) trace[nt] = (orc_trace){K, W, 3, 0};
        nt++;
        polishSz(S, z, d, u, G, g, N, J, tol); /* :366 */
        ret = iter;                            /* :374 */
        break;
    }
    if (ntrace_out) *ntrace_out = nt;
    work_free(w);
    return ret;
}

/* ------------------------------------------------------------------------- */
/* src/Simplex.jl:445-615  cDantzigLP  (bounded simplex, Dantzig -> Bland)    */
/* ------------------------------------------------------------------------- */
/* A is M x N column-major.  B (M, in/out, 0-based, sorted), S (N, in/out),
 * invB (M x M in/out), q (M in/out), x (N out).  Returns status 1/2/3, or
 * -1 when lu() would throw. */
static int cDantzigLP(const double *c, const double *A, const double *b, const double *d,
                      const double *u, int *B, int32_t *S, double *invB, double *q, double *x,
                      int N, int M, double tol)
{
    char *F = xm(N);
    double *gt = xm(sizeof(double) * (M + 1));
    int *ip = xm(sizeof(int) * (M + 1));
    int32_t *Sb = xm(sizeof(int32_t) * (M + 1));
    double *ud = xm(sizeof(double) * N), *cA = xm(sizeof(double) * N);
    double *Y = xm(sizeof(double) * (size_t)M * N); /* M x nF, columns in F order */
    double *h = xm(sizeof(double) * N);
    int *iFl = xm(sizeof(int) * N); /* findall(F) */
    int *iH = xm(sizeof(int) * N);
    double *hp = xm(sizeof(double) * N);
    double *p = xm(sizeof(double) * (M + 1));
    double *AB = xm(sizeof(double) * (size_t)M * M), *wk = xm(sizeof(double) * (size_t)M * M);
    int *piv = xm(sizeof(int) * (M + 1));
    double *ib = xm(sizeof(double) * (M + 1));
    int status = 1;

    for (int k = 0; k < N; ++k) F[k] = 1;
    for (int j = 0; j < M; ++j) F[B[j]] = 0;
    for (int k = 0; k < N; ++k) {
        ud[k] = u[k] - d[k];
        double s = 0.0; /* norm(A[:,k]) :464 */
        for (int r = 0; r < M; ++r) s += A[IDX(r, k, M)] * A[IDX(r, k, M)];
        cA[k] = sqrt(s);
        x[k] = (S[k] == ST_UP) ? u[k] : d[k]; /* :461,:471-472 */
    }
    int nF = 0, nH = 0;
#define RECOMPUTE_Y()                                                                \
    do {                                                                             \
        nF = 0;                                                                      \
        for (int k = 0; k < N; ++k)                                                  \
            if (F[k]) {                                                              \
                for (int r = 0; r < M; ++r) {                                        \
                    double s = 0.0;                                                  \
                    for (int t = 0; t < M; ++t) s += invB[IDX(r, t, M)] * A[IDX(t, k, M)]; \
                    Y[IDX(r, nF, M)] = s;                                            \
                }                                                                    \
                iFl[nF++] = k;                                                       \
            }                                                                        \
    } while (0)
#define RECOMPUTE_H()                                                                \
    do {                                                                             \
        nH = 0;                                                                      \
        for (int f = 0; f < nF; ++f) {                                               \
            int k = iFl[f];                                                          \
            double s = 0.0;                                                          \
            for (int r = 0; r < M; ++r) s += Y[IDX(r, f, M)] * c[B[r]];              \
            double hv = c[k] - s;                                                    \
            if (S[k] == ST_DN) hv = -hv;                                             \
            h[f] = hv;                                                               \
            if (hv > tol) { hp[nH] = hv; iH[nH] = k; nH++; }                         \
        }                                                                            \
    } while (0)

    RECOMPUTE_Y(); /* :475 */
    RECOMPUTE_H(); /* :476-483 */
    int Bland = 0, loop = 0;
    while (nH > 0) { /* :486 */
        loop += 1;
        if (loop > N) Bland = 1;
        int k0 = 0; /* :495 argmax(hp ./ cA[iH]), first maximum */
        if (!Bland) {
            double best = hp[0] / cA[iH[0]];
            for (int t = 1; t < nH; ++t) {
                double v = hp[t] / cA[iH[t]];
                if (v > best) { best = v; k0 = t; }
            }
        }
        int k = iH[k0];
        for (int r = 0; r < M; ++r) { /* p = invB*A[:,k] :497 */
            double s = 0.0;
            for (int t = 0; t < M; ++t) s += invB[IDX(r, t, M)] * A[IDX(t, k, M)];
            p[r] = s;
        }
        int kd = (S[k] == ST_DN);
        int m = 0, l = 0;
        int32_t Sl = ST_DN;
        if (kd) { /* :500-540 */
            for (int j = 0; j < M; ++j) {
                int i = B[j];
                if (p[j] > tol) { gt[m] = (q[j] - d[i]) / p[j]; ip[m] = j; Sb[m] = ST_DN; m++; }
                else if (p[j] < -tol) { gt[m] = (q[j] - u[i]) / p[j]; ip[m] = j; Sb[m] = ST_UP; m++; }
            }
            if (m == 0) {
                if (u[k] < INFINITY) l = -1;
                else { for (int j = 0; j < M; ++j) x[B[j]] = q[j]; status = 3; goto done; }
            } else {
                int li = 0; /* findmin(gt[1:m]) first minimum */
                for (int t = 1; t < m; ++t) if (gt[t] < gt[li]) li = t;
                double gl = gt[li];
                if (u[k] < INFINITY) {
                    if (gl >= ud[k]) l = -1;
                    else { Sl = Sb[li]; l = ip[li] + 1; }
                } else {
                    if (isinf(gl)) { for (int j = 0; j < M; ++j) x[B[j]] = q[j]; status = 3; goto done; }
                    Sl = Sb[li]; l = ip[li] + 1;
                }
            }
        } else { /* UP :542-569 */
            for (int j = 0; j < M; ++j) {
                int i = B[j];
                if (p[j] > tol) { gt[m] = (q[j] - u[i]) / p[j]; ip[m] = j; Sb[m] = ST_UP; m++; }
                else if (p[j] < -tol) { gt[m] = (q[j] - d[i]) / p[j]; ip[m] = j; Sb[m] = ST_DN; m++; }
            }
            if (m == 0) l = -2;
            else {
                int li = 0; /* findmax(gt[1:m]) first maximum */
                for (int t = 1; t < m; ++t) if (gt[t] > gt[li]) li = t;
                double gl = gt[li];
                if (gl <= -ud[k]) l = -2;
                else { Sl = Sb[li]; l = ip[li] + 1; }
            }
        }
        if (l == -1) { S[k] = ST_UP; x[k] = u[k]; }        /* :572-575 */
        else if (l == -2) { S[k] = ST_DN; x[k] = d[k]; }   /* :576-579 */
        else if (l > 0) {                                  /* :580-597 */
            int mrow = l - 1;
            int lv = B[mrow];
            F[k] = 0; F[lv] = 1; B[mrow] = k;
            /* sort!(B) */
            for (int a = 1; a < M; ++a) {
                int v = B[a], t = a - 1;
                while (t >= 0 && B[t] > v) { B[t + 1] = B[t]; t--; }
                B[t + 1] = v;
            }
            for (int j = 0; j < M; ++j)
                for (int r = 0; r < M; ++r) AB[IDX(r, j, M)] = A[IDX(r, B[j], M)];
            if (lu_inverse(AB, M, wk, piv) != 0) { status = -1; goto done; }
            memcpy(invB, AB, sizeof(double) * (size_t)M * M);
            S[k] = ST_IN; S[lv] = Sl;
            x[lv] = (Sl == ST_DN) ? d[lv] : u[lv];
            RECOMPUTE_Y();
        }
        /* q = invB*b - Y*x[F]  :599 */
        for (int r = 0; r < M; ++r) {
            double s = 0.0;
            for (int t = 0; t < M; ++t) s += invB[IDX(r, t, M)] * b[t];
            ib[r] = s;
        }
        for (int r = 0; r < M; ++r) p[r] = 0.0;
        for (int f = 0; f < nF; ++f) {
            double xv = x[iFl[f]];
            if (xv != 0.0)
                for (int r = 0; r < M; ++r) p[r] += Y[IDX(r, f, M)] * xv;
        }
        for (int r = 0; r < M; ++r) q[r] = ib[r] - p[r];
        RECOMPUTE_H(); /* :600-606 */
    }
    for (int j = 0; j < M; ++j) x[B[j]] = q[j]; /* :610 */
    {
        int ms = 0; /* :612-613 */
        for (int f = 0; f < nF; ++f) if (fabs(h[f]) < tol) ms = 1;
        status = ms ? 2 : 1;
    }
done:
    free(F); free(gt); free(ip); free(Sb); free(ud); free(cA); free(Y); free(h); free(iFl);
    free(iH); free(hp); free(p); free(AB); free(wk); free(piv); free(ib);
    return status;
#undef RECOMPUTE_Y
#undef RECOMPUTE_H
}

/*
 * src/SSQP.jl:461-560  initQP(Q, settingsLP): Phase-1 feasible vertex.
 * Outputs x (N), S (N+J); returns 1 (feasible), 0 (infeasible), -1 (lu failed).
 */
int orc_initQP(int N, int M, int J, const double *A, const double *G, const double *b,
               const double *g, const double *d, const double *u, double tol, double *x,
               int32_t *S)
{
    /* free variables and (-inf,u] variables  :485-490 */
    int n = 0, m_id = 0;
    int *iv = xm(sizeof(int) * (N + 1)), *id = xm(sizeof(int) * (N + 1));
    for (int k = 0; k < N; ++k) {
        int fu = (u[k] == INFINITY), fd = (d[k] == -INFINITY);
        if (fu && fd) iv[n++] = k;
        else if (fd) id[m_id++] = k;
    }
    int M0 = M + J, N0 = N + J + n, N1 = M0 + N0;
    double *A1 = calloc((size_t)M0 * N1 + 1, sizeof(double));
    double *b0 = xm(sizeof(double) * (M0 + 1));
    double *d1 = xm(sizeof(double) * N1), *u1 = xm(sizeof(double) * N1), *c1 = xm(sizeof(double) * N1);
    double *x1 = xm(sizeof(double) * N1);
    int32_t *S1 = xm(sizeof(int32_t) * N1);
    int *B = xm(sizeof(int) * (M0 + 1));
    double *invB = calloc((size_t)M0 * M0 + 1, sizeof(double));
    double *q = xm(sizeof(double) * (M0 + 1));
    int ret = 1;
    /* A0 = [A 0 -A[:,iv]; G I -G[:,iv]]   :495-496 */
    for (int k = 0; k < N; ++k) {
        for (int r = 0; r < M; ++r) A1[IDX(r, k, M0)] = A[IDX(r, k, M)];
        for (int r = 0; r < J; ++r) A1[IDX(M + r, k, M0)] = G[IDX(r, k, J)];
    }
    for (int j = 0; j < J; ++j) A1[IDX(M + j, N + j, M0)] = 1.0;
    for (int t = 0; t < n; ++t) {
        int k = iv[t];
        for (int r = 0; r < M; ++r) A1[IDX(r, N + J + t, M0)] = -A[IDX(r, k, M)];
        for (int r = 0; r < J; ++r) A1[IDX(M + r, N + J + t, M0)] = -G[IDX(r, k, J)];
    }
    for (int r = 0; r < M; ++r) b0[r] = b[r];
    for (int r = 0; r < J; ++r) b0[M + r] = g[r];
    for (int k = 0; k < N; ++k) { d1[k] = d[k]; u1[k] = u[k]; }
    for (int k = N; k < N0; ++k) { d1[k] = 0.0; u1[k] = INFINITY; }
    for (int t = 0; t < n; ++t) d1[iv[t]] = 0.0; /* :503 */
    for (int t = 0; t < m_id; ++t) {             /* :507-509 */
        int k = id[t];
        d1[k] = -u1[k];
        u1[k] = INFINITY;
        for (int r = 0; r < M0; ++r) A1[IDX(r, k, M0)] = -A1[IDX(r, k, M0)];
    }
    for (int k = 0; k < N1; ++k) S1[k] = ST_DN; /* :512-514 */
    for (int j = 0; j < M0; ++j) { B[j] = N0 + j; S1[N0 + j] = ST_IN; }
    /* q = A0*d0; invB diag sign; q = abs(q - b0)  :516-521 */
    for (int r = 0; r < M0; ++r) q[r] = 0.0;
    for (int k = 0; k < N0; ++k) {
        double dk = d1[k];
        if (dk != 0.0)
            for (int r = 0; r < M0; ++r) q[r] += A1[IDX(r, k, M0)] * dk;
    }
    for (int j = 0; j < M0; ++j) {
        double sgn = (b0[j] >= q[j]) ? 1.0 : -1.0;
        invB[IDX(j, j, M0)] = sgn;
        A1[IDX(j, N0 + j, M0)] = sgn; /* A1 = [A0 invB] :523 */
        q[j] = fabs(q[j] - b0[j]);
    }
    for (int k = 0; k < N0; ++k) c1[k] = 0.0;
    for (int k = N0; k < N1; ++k) { c1[k] = 1.0; d1[k] = 0.0; u1[k] = INFINITY; }

    int st = cDantzigLP(c1, A1, b0, d1, u1, B, S1, invB, q, x1, N1, M0, tol); /* :530 */
    if (st < 0) { ret = -1; }
    for (int k = 0; k < N; ++k) x[k] = x1[k];      /* :531 */
    for (int k = 0; k < N + J; ++k) S[k] = S1[k];  /* :532 */
    if (ret > 0) {
        double f = 0.0;
        for (int k = N0; k < N1; ++k) f += x1[k];  /* :533 */
        if (f > tol) ret = 0;                      /* :534-537 */
    }
    if (ret > 0) {
        for (int k = N; k < N + J; ++k) S[k] = (S[k] == ST_IN) ? ST_OE : ST_EO; /* :540-542 */
        for (int t = 0; t < n; ++t) {                                            /* :544-547 */
            x[iv[t]] -= x1[N + J + t];
            S[iv[t]] = ST_IN;
        }
        for (int t = 0; t < m_id; ++t) x[id[t]] = -x[id[t]]; /* :551; :552-557 is a no-op */
    }
    free(iv); free(id); free(A1); free(b0); free(d1); free(u1); free(c1); free(x1); free(S1);
    free(B); free(invB); free(q);
    return ret;
}

/*
 * src/SSQP.jl:224-234  solveQP(Q; settings, settingsLP)
 * mc is the model code of the QP constructor (types.jl:240-284); mc <= 0
 * returns (zeros, fill(DN,N), -1) with S of length N only in the reference --
 * here S[0:N] is filled with DN and S[N:N+J] left untouched.
 */
int64_t orc_solveQP(int N, int M, int J, const double *V, const double *A, const double *G,
                    const double *q, const double *b, const double *g, const double *d,
                    const double *u, int mc, int32_t *S, double *z, const orc_settings *st,
                    int32_t *detail, orc_trace *trace, int ntrace, int *ntrace_out)
{
    if (ntrace_out) *ntrace_out = 0;
    if (detail) *detail = ORC_OK;
    if (mc <= 0) {
        for (int k = 0; k < N; ++k) { z[k] = 0.0; S[k] = ST_DN; }
        return -1;
    }
    double *x0 = xm(sizeof(double) * N);
    int s = orc_initQP(N, M, J, A, G, b, g, d, u, st->tol, x0, S);
    if (s <= 0) {
        memcpy(z, x0, sizeof(double) * (size_t)N);
        if (s < 0 && detail) *detail = ORC_SINGULAR_LU;
        free(x0);
        return s;
    }
    int64_t r = orc_solveQP_warm(N, M, J, V, A, G, q, b, g, d, u, S, x0, z, st, detail, trace,
                                 ntrace, ntrace_out);
    free(x0);
    return r;
}

/* Batch driver used by tests and by bench.py's cpu_baseline leg: problems are
 * stored back to back; one QP per OpenMP thread.  Returns threads used. */
int orc_solveQP_warm_batch2(int nprob, int N, int M, int J, const double *V, const double *A,
                            const double *G, const double *q, const double *b, const double *g,
                            const double *d, const double *u, int32_t *S, const double *x0,
                            double *z, const orc_settings *st, int64_t *status, int32_t *detail,
                            int nthreads, int lapack);
int orc_solveQP_warm_batch(int nprob, int N, int M, int J, const double *V, const double *A,
                           const double *G, const double *q, const double *b, const double *g,
                           const double *d, const double *u, int32_t *S, const double *x0,
                           double *z, const orc_settings *st, int64_t *status, int32_t *detail,
                           int nthreads)
{
    return orc_solveQP_warm_batch2(nprob, N, M, J, V, A, G, q, b, g, d, u, S, x0, z, st, status, detail, nthreads, 0);
}

int orc_solveQP_warm_batch3(int nprob, int N, int M, int J, const double *V, const double *A,
                            const double *G, const double *q, const double *b, const double *g,
                            const double *d, const double *u, int32_t *S, const double *x0,
                            double *z, const orc_settings *st, int64_t *status, int32_t *detail,
                            int nthreads, int lapack, double *lambda, double *gamma);
/* lapack != 0: the dense arithmetic by LAPACK/BLAS (orc_lapack_load must have succeeded) */
int orc_solveQP_warm_batch2(int nprob, int N, int M, int J, const double *V, const double *A,
                            const double *G, const double *q, const double *b, const double *g,
                            const double *d, const double *u, int32_t *S, const double *x0,
                            double *z, const orc_settings *st, int64_t *status, int32_t *detail,
                            int nthreads, int lapack)
{
    return orc_solveQP_warm_batch3(nprob, N, M, J, V, A, G, q, b, g, d, u, S, x0, z, st, status, detail, nthreads,
                                   lapack, NULL, NULL);
}

/* lambda (nprob x (M+J)) and gamma (nprob x N) may be NULL: the multipliers of the last pass (orc_solveQP_warm_ex) */
int orc_solveQP_warm_batch3(int nprob, int N, int M, int J, const double *V, const double *A,
                            const double *G, const double *q, const double *b, const double *g,
                            const double *d, const double *u, int32_t *S, const double *x0,
                            double *z, const orc_settings *st, int64_t *status, int32_t *detail,
                            int nthreads, int lapack, double *lambda, double *gamma)
{
    int used = 1;
    if (lapack && !LP.h) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = nthreads > 0 ? nthreads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int p = 0; p < nprob; ++p) {
        size_t P = (size_t)p;
        int32_t det = 0;
        t_lapack = lapack;
        status[p] = orc_solveQP_warm_ex(N, M, J, V + P * N * N, A + P * M * N, G + P * J * N,
                                        q + P * N, b + P * M, g + P * J, d + P * N, u + P * N,
                                        S + P * (N + J), x0 + P * N, z + P * N, st, &det, NULL, 0,
                                        NULL, lambda ? lambda + P * (M + J) : NULL, gamma ? gamma + P * N : NULL);
        if (detail) detail[p] = det;
        t_lapack = 0;
    }
    return used;
}

int orc_initQP_batch(int nprob, int N, int M, int J, const double *A, const double *G,
                     const double *b, const double *g, const double *d, const double *u,
                     double tol, double *x, int32_t *S, int32_t *status, int nthreads)
{
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = nthreads > 0 ? nthreads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int p = 0; p < nprob; ++p) {
        size_t P = (size_t)p;
        status[p] = orc_initQP(N, M, J, A + P * M * N, G + P * J * N, b + P * M, g + P * J,
                               d + P * N, u + P * N, tol, x + P * N, S + P * (N + J));
    }
    return used;
}
