/*
 * ssqp_hip.h -- C ABI of libssqp_hip.so, the MI355X (gfx950) backend for the
 * active-set inner loop of PharosAbad/StatusSwitchingQP.jl.
 *
 * The reference has NO FFI layer: its boundary is the Julia method
 *     solveQP(Q::QP{T}, S, x0; settings)            src/SSQP.jl:237-377
 * and its wrapper
 *     solveQP(Q::QP{T}; settings, settingsLP)       src/SSQP.jl:224-234
 * The entry points below are what a `ccall` from those two methods binds
 * (INTEGRATION.md shows the Julia side).  Conventions, all taken from the
 * reference's data model (src/types.jl):
 *   - matrices are column-major Float64 (Julia Matrix{Float64}):
 *       V  N x N (symmetric, as stored by the QP constructor, types.jl:243)
 *       A  M x N,  G  J x N  (M and J may be 0; the pointer is then ignored)
 *   - vectors q,d,u (N), b (M), g (J); +-Inf encode missing bounds
 *   - S is Vector{Status} reinterpreted as Int32, length N+J, codes
 *       IN=0 DN=1 UP=2 OE=3 EO=4  (types.jl:17-23); mutated in place
 *   - x0 (N) is read only; z (N) receives the solution (SSQP.jl:266)
 *   - *status is the reference's third return value: iter > 0 on success,
 *       0 infeasible (Phase-1), -1 numerical/model error, -(maxIter+1) when the
 *       iteration limit is hit (SSQP.jl:205-209, 272-273)
 *   - *detail says WHY status == -1 where Julia would have thrown
 *       (SSQP_DETAIL_*), 0 otherwise
 * The int return value of every function is an infrastructure code
 * (SSQP_OK / SSQP_ERR_*), never the solver status.
 *
 * The library is re-entrant per ssqp_ctx (one ctx per GPU / per host thread).
 */
#ifndef SSQP_HIP_H
#define SSQP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* infrastructure return codes */
enum {
    SSQP_OK = 0,
    SSQP_ERR_ARG = 1,       /* bad dimension / null pointer */
    SSQP_ERR_NO_DEVICE = 2, /* no HIP device: the product has no CPU fallback */
    SSQP_ERR_HIP = 3,       /* a HIP runtime call failed (ssqp_last_error) */
    SSQP_ERR_ALLOC = 4,
    SSQP_ERR_UNSUPPORTED = 5 /* e.g. settings.rule != Dantzig */
};

/* Status codes of S (src/types.jl:17-23) */
enum { SSQP_IN = 0, SSQP_DN = 1, SSQP_UP = 2, SSQP_OE = 3, SSQP_EO = 4 };

/* detail codes (why status == -1) */
enum {
    SSQP_DETAIL_NONE = 0,
    SSQP_DETAIL_POSDEF_V = 1,  /* cholesky(V[F,F]) not PD    (SSQP.jl:322 throws) */
    SSQP_DETAIL_POSDEF_C = 2,  /* cholesky(A_E V^-1 A_E') not PD (SSQP.jl:328 throws) */
    SSQP_DETAIL_SINGULAR_LU = 3, /* Phase-1 basis singular   (Simplex.jl:590 throws) */
    SSQP_DETAIL_MODEL = 4      /* Q.mc <= 0                  (SSQP.jl:226-228) */
};

/* Settings{Float64} (src/types.jl:390-408).  `pivot` is read nowhere in the
 * reference's live code and is not mirrored; `rule`: 0 = :Dantzig (the only
 * Phase-1 rule implemented; others -> SSQP_ERR_UNSUPPORTED). */
typedef struct ssqp_settings {
    int32_t maxIter; /* 7777 */
    int32_t rule;    /* 0 */
    double tol;      /* 2^-26 */
    double tolG;     /* 2^-33 */
} ssqp_settings;

/* per-problem accounting written by the kernel (for the roofline figures,
 * SURVEY.md section 8(d)): sums over the iterations the problem ran */
typedef struct ssqp_stats {
    int64_t iters;      /* loop passes (== status when status > 0) */
    int64_t alg_bytes;  /* dense formulation: sum_i 8(K^2+RK[+R^2]) + 8(M+J)N + 48N + 4(N+J) */
    int64_t read_bytes; /* this kernel's formulation: sum_i 8N(K + C_i) + 8(M+J)N + 48N + 4(N+J), C_i = columns
                           of V with a nonzero weight in the gamma pass (0 when the step was blocked) */
    int64_t alg_flops;  /* sum_i K^3 + 4K^2W + 2K^2 + 2R^2 + 4RK + ... */
    int64_t sum_k3;     /* sum_i K^3 */
    int32_t max_k;      /* largest free set seen */
    int32_t path;       /* bit0: LDS factor path used, bit1: global-scratch path, bit2: kept-factor engine, bit3: kept
                           factor migrated to the global arena, bit4: wavefront-per-QP kernel, bit5: handed over
                           from the wavefront kernel to the workgroup kernel, bit6: continued in the big-factor build
                           of the wavefront kernel after a hand-over from the build it started in */
} ssqp_stats;

/* one record per loop pass when tracing is requested */
typedef struct ssqp_trace {
    int32_t K;    /* free variables */
    int32_t W;    /* constraint rows kept after the rank filter */
    int32_t kind; /* 0 K==0 pass (freeK!), 1 blocked step (aStep!), 2 release (KKTchk!), 3 optimal */
    int32_t id;   /* smallest switched id, 1-based, inequalities N+j; 0 if none */
} ssqp_trace;

typedef struct ssqp_ctx ssqp_ctx;

/* ---- context ---------------------------------------------------------- */
/* Creates a context on HIP device `device`.  Fails with SSQP_ERR_NO_DEVICE
 * when no GPU is present: there is deliberately no CPU fallback. */
int ssqp_ctx_create(int device, ssqp_ctx **out);
int ssqp_ctx_destroy(ssqp_ctx *ctx);
const char *ssqp_last_error(const ssqp_ctx *ctx);
/* Per-context switches (never read from the environment; a context belongs to one host thread at a time).
 * The reference's Settings (src/types.jl:390-397) stay free of backend fields.
 *   "wave_kernel"     1 (default): QPs of a shape the wavefront-per-QP kernel takes (N even <= 512, M+J <= 11)
 *                     run there: they start in the build "wave_qp_per_cu" selects (factor of up to 92 / 127 rows), a QP
 *                     whose free set outgrows that continues in the big-factor build (four row slots, up to 252 free
 *                     variables, rows >= 64 of the factor in global scratch) and only beyond that in the workgroup
 *                     kernel -- every hand-over passes (z, S, pass count), the loop has no other state;
 *                     2: start in the big-factor build (workloads known to end with large free sets);
 *                     0: workgroup kernel only
 *   "wave_qp_per_cu"  QPs (wavefronts) per CU of that kernel: 4 = one per SIMD, 512 registers, the factor (up to 92 rows)
 *                     in LDS -- the lowest latency per QP; 8 = two per SIMD, 256 registers, rows >= 64 of the factor (and,
 *                     between the passes that use it, the second row slot) in global scratch, up to 127 rows -- more
 *                     QPs per second when more than 4 per CU are in flight (large batches,
 *                     several contexts on different streams); 0 (default) = 8 for batches above 4 * numCU QPs, else 4
 *   "incremental"     1 (default): keep the LDL' factor of V[F,F] across passes; 0: refactor in every pass like
 *                     SSQP.jl:322 (workgroup kernel)
 *   "dense_gamma"     1: the reference's dense formulation (from-scratch factor, gamma pass over all N columns,
 *                     SSQP.jl:322,352) -- the HBM roofline measurement; default 0
 *   "wg_per_cu"       workgroups per CU of the workgroup kernel: 0 = automatic (default), 1, 2
 *   "phase1_wave"     1 (default): ssqp_phase1_batch_dev_f64 runs one WAVEFRONT per QP where that kernel applies (M + J <= 11,
 *                     N + J + M + J <= 576; QPs with free variables are left to the workgroup kernel); 0: workgroup kernel only
 *   "lazy_handover"   1: the launch of the workgroup kernel on the wavefront kernel's hand-over list is not queued
 *                     behind it but issued by ssqp_sync (or the next call on the context) and only when the list is
 *                     not empty -- for hosts that keep several contexts busy on different streams (the queued launch
 *                     would wait for a free CU even when it has nothing to do).  The results of a device-buffer call
 *                     are then complete only after ssqp_sync(ctx, stream): do NOT consume them stream-ordered.
 *                     Default 0 (everything is queued on the caller's stream)
 *   "pin_host_buffers" 1: ssqp_solve_batch_f64 page-locks the caller's V array in place (hipHostRegister) and keeps it
 *                     registered until it is called with another array, the option is set back to 0 (that call
 *                     unregisters) or the context is destroyed: uploads at the full PCIe rate for hosts that solve out
 *                     of the same buffers repeatedly.  LIFETIME: the array must stay allocated, at the same address,
 *                     for as long as it is registered -- set the option to 0 (or destroy the context) BEFORE freeing
 *                     it; a registration is recognised by (pointer, size) only.  Default 0
 * Unknown names / out-of-range values: SSQP_ERR_ARG. */
int ssqp_ctx_set_option(ssqp_ctx *ctx, const char *name, int value);
int ssqp_ctx_get_option(ssqp_ctx *ctx, const char *name, int *value);
/* library/version string, safe without a GPU */
const char *ssqp_version(void);
/* Settings{Float64}() defaults (types.jl:401-408) */
void ssqp_default_settings(ssqp_settings *s);

/* ---- hot path: replaces solveQP(Q, S, x0; settings), SSQP.jl:237 -------- */
/* Host buffers (what the Julia ccall passes).  Copies the problem to the
 * device, runs the in-kernel active-set loop, copies z,S back. */
int ssqp_solve_f64(ssqp_ctx *ctx, int N, int M, int J, const double *V, const double *A,
                   const double *G, const double *q, const double *b, const double *g,
                   const double *d, const double *u, int32_t *S, const double *x0, double *z,
                   const ssqp_settings *settings, int64_t *status, int32_t *detail);

/* ---- replaces solveQP(Q; settings, settingsLP), SSQP.jl:224 ------------- */
/* mc = Q.mc (types.jl:240-284).  Phase-1 (initQP, SSQP.jl:461-560) runs on
 * the host, the loop on the GPU. */
int ssqp_solve_full_f64(ssqp_ctx *ctx, int N, int M, int J, const double *V, const double *A,
                        const double *G, const double *q, const double *b, const double *g,
                        const double *d, const double *u, int mc, int32_t *S, double *z,
                        const ssqp_settings *settings, const ssqp_settings *settingsLP,
                        int64_t *status, int32_t *detail);

/* ---- batches of independent QPs of equal shape --------------------------- */
/* Host buffers, problems stored back to back (problem p at offset p*len).  Large batches go up in chunks of about
 * 256 MiB of V and every chunk is solved on one of four internal launch lanes as soon as it has landed: the call takes
 * the PCIe transfer plus one chunk's solve. */
/* lambda (nprob x (M+J)) and gamma (nprob x N), both optional (NULL = not wanted): the Lagrange multipliers of the
 * LAST pass of every QP with status > 0, laid out by constraint row / variable id --
 *   lambda[r]  alphaL (SSQP.jl:351) of row r of [A;G] when the row was in the working set and kept by the rank
 *              filter; for an active inequality the filter purged, the value KKTchk! computes for it (SSQP.jl:158-159);
 *              0 for inactive inequalities and purged equality rows (the reference computes nothing for those)
 *   gamma[i]   gamma (SSQP.jl:352) of bound variable i (sign as in the reference: an UP variable is optimal when
 *              gamma <= tolG, a DN variable when gamma >= -tolG); 0 for free variables
 * On the K == 0 exit (SSQP.jl:278-285; no multipliers exist there) gamma = V z + q, lambda = 0.  Not written for
 * status <= 0: every entry point (host buffers and device buffers alike) ZEROES the arrays it is given before the launch
 * -- on `stream` for the device entries -- so a QP that does not converge reads 0, never an earlier call's values.  The reference keeps these inside solveQP; they are exported so that north_star's "x/lambda within
 * 1e-10" can be checked at the boundary. */
int ssqp_solve_batch_f64(ssqp_ctx *ctx, int nprob, int N, int M, int J, const double *V,
                         const double *A, const double *G, const double *q, const double *b,
                         const double *g, const double *d, const double *u, int32_t *S,
                         const double *x0, double *z, const ssqp_settings *settings,
                         int64_t *status, int32_t *detail, ssqp_stats *stats, double *lambda, double *gamma);

/* A batch KEPT in HBM: upload the problem data once, solve it any number of times -- warm starts from another
 * (S, x0) (the three-argument solveQP, SSQP.jl:237), efficient-frontier sweeps that only replace q or b
 * (QP(P, q, L) / QP(P, mu, q), types.jl:303-339).  ssqp_problem_solve moves only S, x0 in and z, S, status out
 * (about 6 KB per N = 512 problem instead of 2 MiB of V).  `which` of ssqp_problem_set_vector: 0 q, 1 b, 2 g,
 * 3 d, 4 u (arrays of nprob * length, problems back to back). */
typedef struct ssqp_problem ssqp_problem;
int ssqp_problem_upload(ssqp_ctx *ctx, int nprob, int N, int M, int J, const double *V, const double *A,
                        const double *G, const double *q, const double *b, const double *g, const double *d,
                        const double *u, ssqp_problem **out);
int ssqp_problem_set_vector(ssqp_problem *p, int which, const double *data);
int ssqp_problem_solve(ssqp_problem *p, int32_t *S, const double *x0, double *z, const ssqp_settings *settings,
                       int64_t *status, int32_t *detail, ssqp_stats *stats);
int ssqp_problem_free(ssqp_problem *p);

/* The same over SEVERAL contexts -- normally one per GPU of the node (ssqp_ctx_create(dev, ...) for dev = 0..G-1):
 * the batch is cut into contiguous blocks, context r solves problems [r*ceil(P/G), (r+1)*ceil(P/G)) on its own
 * GPU from its own host thread, and every block's z, S, status land in the caller's arrays (SURVEY.md section 8e:
 * independent QPs shard with no exchange; in a one-process host this call IS the final gather).  What a Julia
 * host binds to solve a Vector{QP} on all GPUs with one ccall.  Two contexts may also share a device. */
int ssqp_solve_batch_multi_f64(ssqp_ctx *const *ctxs, int nctx, int nprob, int N, int M, int J,
                               const double *V, const double *A, const double *G, const double *q,
                               const double *b, const double *g, const double *d, const double *u,
                               int32_t *S, const double *x0, double *z, const ssqp_settings *settings,
                               int64_t *status, int32_t *detail, ssqp_stats *stats);

/* Device-resident buffers (all pointers are device pointers on ctx's GPU;
 * stats/trace and the multiplier outputs dlambda/dgamma -- see ssqp_solve_batch_f64 -- may be NULL).  Asynchronous on `stream` (a hipStream_t passed
 * as void*; NULL = HIP's default stream); ssqp_sync waits for it.
 * trace holds ntrace records per problem. */
int ssqp_solve_batch_dev_f64(ssqp_ctx *ctx, int nprob, int N, int M, int J, const double *dV,
                             const double *dA, const double *dG, const double *dq,
                             const double *db, const double *dg, const double *dd,
                             const double *du, int32_t *dS, const double *dx0, double *dz,
                             const ssqp_settings *settings, int64_t *dstatus, int32_t *ddetail,
                             ssqp_stats *dstats, ssqp_trace *dtrace, int ntrace, double *dlambda, double *dgamma,
                             void *stream);
/* The same with per-array problem strides in ELEMENTS (NULL = dense batch as above; a stride of 0 = that array is
 * shared by every problem).  Efficient-frontier style batches -- QP(P, q, L) / QP(P, mu, q) of the reference
 * (src/types.jl:303-339): one V (and A, G, d, u) for many q or b -- then read V out of L2 / Infinity Cache. */
typedef struct ssqp_batch_strides {
    size_t V, A, G, q, b, g, d, u;
} ssqp_batch_strides;
int ssqp_solve_batch_strided_dev_f64(ssqp_ctx *ctx, int nprob, int N, int M, int J, const double *dV,
                                     const double *dA, const double *dG, const double *dq,
                                     const double *db, const double *dg, const double *dd,
                                     const double *du, const ssqp_batch_strides *strides, int32_t *dS,
                                     const double *dx0, double *dz, const ssqp_settings *settings,
                                     int64_t *dstatus, int32_t *ddetail, ssqp_stats *dstats,
                                     ssqp_trace *dtrace, int ntrace, double *dlambda, double *dgamma, void *stream);
/* solveQP(Q::QP; settings, settingsLP) for a batch resident in HBM, BOTH stages in ONE kernel launch per QP
 * (replaces SSQP.jl:224-234 = initQP, :229, + the loop, :233): the wavefront that finds a QP's Phase-1 vertex goes straight on
 * into the active-set loop for it -- no (x0, S0) round trip, no second launch.  dS and dz are outputs only.  Per QP the
 * result is the reference's triple: status > 0 (passes) with (z, S) of the loop, or Phase-1's verdict (0 infeasible, -1
 * singular basis with detail SSQP_DETAIL_SINGULAR_LU) with z = x0 and Phase-1's S, as SSQP.jl:230-232 returns them.
 * settingsLP NULL = settings (the reference's default).  The model check `Q.mc <= 0` (SSQP.jl:226-228) belongs to the QP
 * constructor and stays with the caller.  Shapes: per-problem arrays, N even <= 512, 1 <= M + J <= 11,
 * N + J + M + J <= 576, default algorithm options (SSQP_ERR_UNSUPPORTED otherwise: call ssqp_phase1_batch_dev_f64 and
 * ssqp_solve_batch_dev_f64).  QPs with free variables take the workgroup Phase-1 kernel behind the same call.
 * Asynchronous on `stream`; lambda / gamma / stats as for ssqp_solve_batch_dev_f64. */
int ssqp_solve_full_batch_dev_f64(ssqp_ctx *ctx, int nprob, int N, int M, int J, const double *dV, const double *dA,
                                  const double *dG, const double *dq, const double *db, const double *dg,
                                  const double *dd, const double *du, int32_t *dS, double *dz,
                                  const ssqp_settings *settings, const ssqp_settings *settingsLP, int64_t *dstatus,
                                  int32_t *ddetail, ssqp_stats *dstats, double *dlambda, double *dgamma, void *stream);
/* "lazy_handover" only: issues the launch the last call on ctx may still owe (waits for that call's wavefront kernel,
 * not for the stream).  A host that reuses the call's in/out buffers (S is in/out) must flush BEFORE it queues work
 * that overwrites them. */
int ssqp_flush(ssqp_ctx *ctx);
/* ssqp_flush, after which `stream` also waits for the owed launch when that went out on ANOTHER stream: what a host
 * calls before it queues, on `stream`, work that overwrites the in/out buffers of the previous call on ctx (a copy
 * that resets S, say) -- ssqp_flush alone orders such work only on the stream of the previous call. */
int ssqp_flush_to(ssqp_ctx *ctx, void *stream);
/* waits for `stream`; with "lazy_handover" it first issues the launch the last call on ctx may still owe */
int ssqp_sync(ssqp_ctx *ctx, void *stream);
/* duration in ms of the solve kernel(s) of the most recent ssqp_solve_batch_dev_f64
 * on this ctx, from HIP events recorded on the launch stream (call after sync) */
int ssqp_last_kernel_ms(ssqp_ctx *ctx, float *ms);
/* the same for the launch `back` launches before the most recent one on this ctx (0 = the most recent; the library keeps
 * the event pairs of the last 16 launches): how a host reads the durations of launches it issued back to back */
int ssqp_recent_kernel_ms(ssqp_ctx *ctx, int back, float *ms);

/* ---- Phase-1: replaces initQP(Q, settingsLP), SSQP.jl:461-560 (host C++) -- */
/* *status: 1 feasible, 0 infeasible, -1 basis singular. */
int ssqp_phase1_f64(int N, int M, int J, const double *A, const double *G, const double *b,
                    const double *g, const double *d, const double *u,
                    const ssqp_settings *settingsLP, double *x0, int32_t *S, int32_t *status);
int ssqp_phase1_batch_f64(int nprob, int N, int M, int J, const double *A, const double *G,
                          const double *b, const double *g, const double *d, const double *u,
                          const ssqp_settings *settingsLP, double *x0, int32_t *S,
                          int32_t *status, int nthreads);

/* The same for a batch whose problem data already sits in HBM: ON the GPU (one wavefront per QP for M + J <= 11 and
 * N + J + M + J <= 576 -- every column of the LP in the registers of its lane --, one workgroup per QP otherwise and for
 * QPs with free variables; option "phase1_wave"), asynchronous on
 * `stream`, bit-identical to the host version (same decisions, same summation orders, no FMA contraction), so the loop
 * that follows runs the same passes.  All pointers are device pointers; needs M + J small enough for the basis
 * inverse to fit in LDS (M + J <= 87; SSQP_ERR_UNSUPPORTED otherwise: use the host version).  dstatus as above.
 * Above 12 rows the kernel (512 threads per QP) leaves out products with an exactly zero entry of inv(B) or of the LU
 * factors -- a simplex basis is mostly unit columns --, which changes no bit for FINITE data; an Inf or NaN in A / G, which
 * the host stage carries through such products (0 * Inf = NaN), may then give a different vertex or status. */
int ssqp_phase1_batch_dev_f64(ssqp_ctx *ctx, int nprob, int N, int M, int J, const double *dA, const double *dG,
                              const double *db, const double *dg, const double *dd, const double *du,
                              const ssqp_settings *settingsLP, double *dx0, int32_t *dS, int32_t *dstatus,
                              void *stream);

/* ---- deterministic synthetic problems ---------- */
/* V = X'X/T + delta*I, X (T x N) iid U(-1/2,1/2) from a SplitMix64 counter
 * stream; A row 0 = ones (budget), rows 1.. = random 0/1 group sums with
 * b = A*(1/N); G iid U(0,1), g_j = gscale*mean(G[j,:]); d = 0, u = ub
 * (+Inf when ub <= 0); q = -qscale*mu, mu iid U(0,0.2).  Sequential f64
 * accumulation without FMA contraction: every caller gets identical bits. */
typedef struct ssqp_gen_cfg {
    int32_t N, M, J, T;
    double delta, ub, gscale, qscale;
} ssqp_gen_cfg;
int ssqp_generate_problem(const ssqp_gen_cfg *cfg, uint64_t seed, double *V, double *A,
                          double *G, double *q, double *b, double *g, double *d, double *u);
int ssqp_generate_batch(const ssqp_gen_cfg *cfg, uint64_t seed0, int nprob, double *V,
                        double *A, double *G, double *q, double *b, double *g, double *d,
                        double *u, int nthreads);
/* V may be NULL in the two calls above (only the small arrays are generated); the N x N x T part can
 * then be produced on the GPU, bit-identical to the host version, straight into device memory: */
int ssqp_generate_V_dev(ssqp_ctx *ctx, const ssqp_gen_cfg *cfg, uint64_t seed0, int nprob, double *dV,
                        void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SSQP_HIP_H */
