// ssqp_host.cpp -- host-side parts of libssqp_hip.so that never touch the GPU:
//   * the deterministic synthetic-problem generator
//   * Phase-1 (initQP + cDantzigLP), the step before the hot path
//     (reference: src/SSQP.jl:461-560, src/Simplex.jl:445-615)
// Compiled by hipcc together with the kernels, but plain C++17.
#include "ssqp_hip.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------ generator
// SplitMix64 used as a counter RNG: value i of stream s of problem `seed` is
// mix(base(seed, s) + (i+1)*GOLDEN); random access, order independent.
constexpr uint64_t GOLDEN = 0x9E3779B97F4A7C15ull;

inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t stream_base(uint64_t seed, uint64_t stream) {
    return mix64(mix64(seed + GOLDEN) ^ (stream * 0xD1B54A32D192ED03ull + GOLDEN));
}
inline double u01(uint64_t base, uint64_t i) {
    return (double)(mix64(base + (i + 1) * GOLDEN) >> 11) * 0x1.0p-53;
}

enum : uint64_t { STREAM_X = 1, STREAM_MU = 2, STREAM_G = 3, STREAM_A = 4 };

int generate_one(const ssqp_gen_cfg &c, uint64_t seed, double *V, double *A, double *G, double *q,
                 double *b, double *g, double *d, double *u) {
    const int N = c.N, M = c.M, J = c.J, T = c.T;
    if (N <= 0 || M < 0 || J < 0 || T <= 0) return SSQP_ERR_ARG;
    const double inf = std::numeric_limits<double>::infinity();
    // V = X'X/T + delta*I; X[t,i] = u01(t + T*i) - 1/2.  The sum over t runs in
    // increasing t for every (i,j): rank-1 accumulation, row t at a time.
    if (V) {
        // blocks of TB sample rows per sweep over V; per element the products are still added
        // one at a time in increasing t (bit-identical to the unblocked rank-1 accumulation)
        constexpr int TB = 16;
        std::vector<double> xt((size_t)TB * N);
        const uint64_t bx = stream_base(seed, STREAM_X);
        std::fill(V, V + (size_t)N * N, 0.0);
        for (int t0 = 0; t0 < T; t0 += TB) {
            const int tb = std::min(TB, T - t0);
            for (int tt = 0; tt < tb; ++tt)
                for (int i = 0; i < N; ++i)
                    xt[(size_t)tt * N + i] = u01(bx, (uint64_t)(t0 + tt) + (uint64_t)T * i) - 0.5;
            const double *__restrict__ xb = xt.data();
            for (int j = 0; j < N; ++j) {
                double *__restrict__ col = V + (size_t)j * N;
                double xj[TB];
                for (int tt = 0; tt < tb; ++tt) xj[tt] = xb[(size_t)tt * N + j];
                for (int i = 0; i <= j; ++i) {  // upper triangle; vectorises over i
                    double s = col[i];
                    for (int tt = 0; tt < tb; ++tt) s += xb[(size_t)tt * N + i] * xj[tt];
                    col[i] = s;
                }
            }
        }
        for (int j = 0; j < N; ++j) {
            double *col = V + (size_t)j * N;
            for (int i = 0; i <= j; ++i) col[i] = col[i] / (double)T;
            col[j] += c.delta;
        }
        for (int j = 0; j < N; ++j)
            for (int i = 0; i < j; ++i) V[(size_t)i * N + j] = V[(size_t)j * N + i];
    }
    {
        const uint64_t bm = stream_base(seed, STREAM_MU);
        for (int i = 0; i < N; ++i) {
            const double mu = 0.2 * u01(bm, i);
            q[i] = (c.qscale == 0.0) ? 0.0 : -c.qscale * mu;
            d[i] = 0.0;
            u[i] = c.ub > 0.0 ? c.ub : inf;
        }
    }
    {
        const uint64_t ba = stream_base(seed, STREAM_A);
        for (int r = 0; r < M; ++r) {
            double cnt = 0.0;
            for (int i = 0; i < N; ++i) {
                double a = 1.0;
                if (r > 0) a = u01(ba, (uint64_t)(r - 1) * N + i) < 0.5 ? 1.0 : 0.0;
                A[(size_t)i * M + r] = a;
                cnt += a;
            }
            b[r] = (r == 0) ? 1.0 : cnt / (double)N;
        }
    }
    {
        const uint64_t bg = stream_base(seed, STREAM_G);
        for (int r = 0; r < J; ++r) {
            double s = 0.0;
            for (int i = 0; i < N; ++i) {
                const double v = u01(bg, (uint64_t)r * N + i);
                G[(size_t)i * J + r] = v;
                s += v;
            }
            g[r] = c.gscale * (s / (double)N);
        }
    }
    return SSQP_OK;
}

template <class Fn>
void parallel_for(int n, int nthreads, Fn fn) {
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    nthreads = std::max(1, std::min(nthreads, n));
    if (nthreads == 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<int> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; ++t)
        pool.emplace_back([&] {
            for (;;) {
                int i = next.fetch_add(1);
                if (i >= n) break;
                fn(i);
            }
        });
    for (auto &th : pool) th.join();
}

// ------------------------------------------------------------------- Phase-1
// Bounded-variable primal simplex with an explicit basis inverse, the combined
// largest-distance/Dantzig rule and the switch to Bland's rule after N loops
// (src/Simplex.jl:445-615), on the slack/artificial LP that initQP builds
// (src/SSQP.jl:484-526).  Decisions (argmax, ratio test, bound flips) follow
// the reference so that (x0, S0) is the vertex the reference would start from.
struct Dense {
    int m = 0, n = 0;
    std::vector<double> a;  // column-major
    Dense() = default;
    Dense(int m_, int n_) : m(m_), n(n_), a((size_t)m_ * n_, 0.0) {}
    double &operator()(int i, int j) { return a[(size_t)j * m + i]; }
    double operator()(int i, int j) const { return a[(size_t)j * m + i]; }
    const double *col(int j) const { return a.data() + (size_t)j * m; }
};

// inverse through LU with partial pivoting (what inv(lu(.)) does); false if singular
bool invert_lu(Dense &a) {
    const int n = a.m;
    std::vector<int> piv(n);
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(a(k, k));
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(a(i, k)) > best) best = std::fabs(a(i, k)), p = i;
        piv[k] = p;
        if (best == 0.0) return false;
        if (p != k)
            for (int j = 0; j < n; ++j) std::swap(a(k, j), a(p, j));
        const double r = 1.0 / a(k, k);
        for (int i = k + 1; i < n; ++i) a(i, k) *= r;
        for (int j = k + 1; j < n; ++j) {
            const double t = a(k, j);
            for (int i = k + 1; i < n; ++i) a(i, j) -= a(i, k) * t;
        }
    }
    Dense x(n, n);
    for (int c = 0; c < n; ++c) {
        double *xc = &x(0, c);
        xc[c] = 1.0;
        for (int k = 0; k < n; ++k)
            if (piv[k] != k) std::swap(xc[k], xc[piv[k]]);
        for (int k = 0; k < n; ++k) {
            const double t = xc[k];
            if (t != 0.0)
                for (int i = k + 1; i < n; ++i) xc[i] -= a(i, k) * t;
        }
        for (int k = n - 1; k >= 0; --k) {
            xc[k] /= a(k, k);
            const double t = xc[k];
            for (int i = 0; i < k; ++i) xc[i] -= a(i, k) * t;
        }
    }
    a = std::move(x);
    return true;
}

struct BoundedSimplex {
    int N, M;
    const Dense &A;
    const std::vector<double> &c, &b, &lo, &hi;
    std::vector<int> &basis;      // sorted
    std::vector<int32_t> &S;      // IN / DN / UP
    Dense invB;
    std::vector<double> xb;       // values of the basic variables (reference's q)
    std::vector<double> x;
    double tol;

    std::vector<char> nonbasic;
    std::vector<int> nb;          // findall(nonbasic)
    Dense Y;                      // invB * A[:, nb]
    std::vector<double> h;        // signed reduced costs over nb
    std::vector<int> cand;        // improving candidates (indices into variables)
    std::vector<double> candH;

    BoundedSimplex(const Dense &A_, const std::vector<double> &c_, const std::vector<double> &b_,
                   const std::vector<double> &lo_, const std::vector<double> &hi_,
                   std::vector<int> &basis_, std::vector<int32_t> &S_, Dense invB_,
                   std::vector<double> xb_, double tol_)
        : N(A_.n), M(A_.m), A(A_), c(c_), b(b_), lo(lo_), hi(hi_), basis(basis_), S(S_),
          invB(std::move(invB_)), xb(std::move(xb_)), tol(tol_) {}

    void refreshY() {
        nb.clear();
        for (int k = 0; k < N; ++k)
            if (nonbasic[k]) nb.push_back(k);
        Y = Dense(M, (int)nb.size());
        for (size_t f = 0; f < nb.size(); ++f) {
            const double *ak = A.col(nb[f]);
            for (int r = 0; r < M; ++r) {
                double s = 0.0;
                for (int t = 0; t < M; ++t) s += invB(r, t) * ak[t];
                Y(r, (int)f) = s;
            }
        }
    }
    void price() {
        cand.clear();
        candH.clear();
        h.assign(nb.size(), 0.0);
        for (size_t f = 0; f < nb.size(); ++f) {
            const int k = nb[f];
            double s = 0.0;
            for (int r = 0; r < M; ++r) s += Y(r, (int)f) * c[basis[r]];
            double hv = c[k] - s;
            if (S[k] == SSQP_DN) hv = -hv;
            h[f] = hv;
            if (hv > tol) cand.push_back(k), candH.push_back(hv);
        }
    }
    void finish() {
        for (int j = 0; j < M; ++j) x[basis[j]] = xb[j];
    }
    // returns 1 unique, 2 alternative optima, 3 unbounded, -1 singular basis
    int run() {
        const double inf = std::numeric_limits<double>::infinity();
        nonbasic.assign(N, 1);
        for (int v : basis) nonbasic[v] = 0;
        std::vector<double> colnorm(N), range(N);
        x.resize(N);
        for (int k = 0; k < N; ++k) {
            range[k] = hi[k] - lo[k];
            double s = 0.0;
            for (int r = 0; r < M; ++r) s += A(r, k) * A(r, k);
            colnorm[k] = std::sqrt(s);
            x[k] = S[k] == SSQP_UP ? hi[k] : lo[k];
        }
        refreshY();
        price();
        std::vector<double> p(M), ratio(M);
        std::vector<int> row(M);
        std::vector<int32_t> leaveTo(M);
        long loop = 0;
        while (!cand.empty()) {
            const bool bland = ++loop > N;
            size_t pick = 0;
            if (!bland) {
                double best = candH[0] / colnorm[cand[0]];
                for (size_t t = 1; t < cand.size(); ++t) {
                    const double v = candH[t] / colnorm[cand[t]];
                    if (v > best) best = v, pick = t;
                }
            }
            const int k = cand[pick];
            const double *ak = A.col(k);
            for (int r = 0; r < M; ++r) {
                double s = 0.0;
                for (int t = 0; t < M; ++t) s += invB(r, t) * ak[t];
                p[r] = s;
            }
            const bool fromLower = S[k] == SSQP_DN;
            int m = 0;
            for (int j = 0; j < M; ++j) {
                const int i = basis[j];
                const bool pos = p[j] > tol, neg = p[j] < -tol;
                if (!pos && !neg) continue;
                // entering from its lower bound moves basic j down when p>0
                const bool toLower = fromLower ? pos : neg;
                ratio[m] = (xb[j] - (toLower ? lo[i] : hi[i])) / p[j];
                row[m] = j;
                leaveTo[m] = toLower ? SSQP_DN : SSQP_UP;
                ++m;
            }
            int action = 0;  // >0: basis row+1 leaves, -1 flip to UP, -2 flip to DN
            int32_t leaveStatus = SSQP_DN;
            if (fromLower) {
                const bool finiteUp = hi[k] < inf;
                if (m == 0) {
                    if (!finiteUp) { finish(); return 3; }
                    action = -1;
                } else {
                    int li = 0;
                    for (int t = 1; t < m; ++t)
                        if (ratio[t] < ratio[li]) li = t;
                    const double gl = ratio[li];
                    if (finiteUp && gl >= range[k]) action = -1;
                    else {
                        if (!finiteUp && std::isinf(gl)) { finish(); return 3; }
                        action = row[li] + 1;
                        leaveStatus = leaveTo[li];
                    }
                }
            } else {
                if (m == 0) action = -2;
                else {
                    int li = 0;
                    for (int t = 1; t < m; ++t)
                        if (ratio[t] > ratio[li]) li = t;
                    if (ratio[li] <= -range[k]) action = -2;
                    else action = row[li] + 1, leaveStatus = leaveTo[li];
                }
            }
            if (action == -1) S[k] = SSQP_UP, x[k] = hi[k];
            else if (action == -2) S[k] = SSQP_DN, x[k] = lo[k];
            else {
                const int leaving = basis[action - 1];
                nonbasic[k] = 0;
                nonbasic[leaving] = 1;
                basis[action - 1] = k;
                std::sort(basis.begin(), basis.end());
                Dense Bm(M, M);
                for (int j = 0; j < M; ++j) std::memcpy(&Bm(0, j), A.col(basis[j]), sizeof(double) * M);
                if (!invert_lu(Bm)) return -1;
                invB = std::move(Bm);
                S[k] = SSQP_IN;
                S[leaving] = leaveStatus;
                x[leaving] = leaveStatus == SSQP_DN ? lo[leaving] : hi[leaving];
                refreshY();
            }
            // xb = invB*b - Y*x[nonbasic]
            std::vector<double> acc(M, 0.0);
            for (size_t f = 0; f < nb.size(); ++f) {
                const double xv = x[nb[f]];
                if (xv != 0.0)
                    for (int r = 0; r < M; ++r) acc[r] += Y(r, (int)f) * xv;
            }
            for (int r = 0; r < M; ++r) {
                double s = 0.0;
                for (int t = 0; t < M; ++t) s += invB(r, t) * b[t];
                xb[r] = s - acc[r];
            }
            price();
        }
        finish();
        for (double hv : h)
            if (std::fabs(hv) < tol) return 2;
        return 1;
    }
};

int phase1_one(int N, int M, int J, const double *A, const double *G, const double *b,
               const double *g, const double *d, const double *u, double tol, double *x0,
               int32_t *S) {
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<int> freeVars, upperOnly;
    for (int k = 0; k < N; ++k) {
        const bool noUp = u[k] == inf, noLo = d[k] == -inf;
        if (noUp && noLo) freeVars.push_back(k);
        else if (noLo) upperOnly.push_back(k);
    }
    const int n = (int)freeVars.size();
    const int M0 = M + J, N0 = N + J + n, N1 = N0 + M0;
    Dense A1(M0, N1);
    std::vector<double> rhs(M0), lo(N1, 0.0), hi(N1, inf), cost(N1, 0.0);
    for (int k = 0; k < N; ++k) {
        for (int r = 0; r < M; ++r) A1(r, k) = A[(size_t)k * M + r];
        for (int r = 0; r < J; ++r) A1(M + r, k) = G[(size_t)k * J + r];
        lo[k] = d[k];
        hi[k] = u[k];
    }
    for (int j = 0; j < J; ++j) A1(M + j, N + j) = 1.0;
    for (int t = 0; t < n; ++t) {
        const int k = freeVars[t];
        for (int r = 0; r < M0; ++r) A1(r, N + J + t) = -A1(r, k);
        lo[k] = 0.0;
    }
    for (int k : upperOnly) {
        lo[k] = -hi[k];
        hi[k] = inf;
        for (int r = 0; r < M0; ++r) A1(r, k) = -A1(r, k);
    }
    for (int r = 0; r < M; ++r) rhs[r] = b[r];
    for (int r = 0; r < J; ++r) rhs[M + r] = g[r];
    std::vector<double> start(M0, 0.0);
    for (int k = 0; k < N0; ++k)
        if (lo[k] != 0.0)
            for (int r = 0; r < M0; ++r) start[r] += A1(r, k) * lo[k];
    Dense invB(M0, M0);
    std::vector<int> basis(M0);
    std::vector<int32_t> S1(N1, SSQP_DN);
    for (int j = 0; j < M0; ++j) {
        const double sgn = rhs[j] >= start[j] ? 1.0 : -1.0;
        invB(j, j) = sgn;
        A1(j, N0 + j) = sgn;
        start[j] = std::fabs(start[j] - rhs[j]);
        basis[j] = N0 + j;
        S1[N0 + j] = SSQP_IN;
        cost[N0 + j] = 1.0;
    }
    BoundedSimplex lp(A1, cost, rhs, lo, hi, basis, S1, invB, start, tol);
    const int st = lp.run();
    for (int k = 0; k < N; ++k) x0[k] = lp.x.empty() ? 0.0 : lp.x[k];
    for (int k = 0; k < N + J; ++k) S[k] = S1[k];
    if (st < 0) return -1;
    double art = 0.0;
    for (int k = N0; k < N1; ++k) art += lp.x[k];
    if (art > tol) return 0;
    for (int k = N; k < N + J; ++k) S[k] = S[k] == SSQP_IN ? SSQP_OE : SSQP_EO;
    for (int t = 0; t < n; ++t) {
        x0[freeVars[t]] -= lp.x[N + J + t];
        S[freeVars[t]] = SSQP_IN;
    }
    for (int k : upperOnly) x0[k] = -x0[k];  // statuses are left as they are (SSQP.jl:552-557 is a no-op)
    return 1;
}

}  // namespace

extern "C" {

#ifndef SSQP_SRC_HASH
#define SSQP_SRC_HASH "unknown"
#endif
// "ssqp_hip <version> (gfx950) src=<sha256 of the sources this binary was built from>" (csrc/Makefile: HASHED)
const char *ssqp_version(void) { return "ssqp_hip 0.3 (gfx950) src=" SSQP_SRC_HASH; }

void ssqp_default_settings(ssqp_settings *s) {
    if (!s) return;
    s->maxIter = 7777;
    s->rule = 0;
    s->tol = 0x1.0p-26;
    s->tolG = 0x1.0p-33;
}

int ssqp_generate_problem(const ssqp_gen_cfg *cfg, uint64_t seed, double *V, double *A, double *G,
                          double *q, double *b, double *g, double *d, double *u) {
    if (!cfg || !q || !d || !u) return SSQP_ERR_ARG;  // V may be NULL: see ssqp_generate_V_dev
    if ((cfg->M > 0 && (!A || !b)) || (cfg->J > 0 && (!G || !g))) return SSQP_ERR_ARG;
    return generate_one(*cfg, seed, V, A, G, q, b, g, d, u);
}

int ssqp_generate_batch(const ssqp_gen_cfg *cfg, uint64_t seed0, int nprob, double *V, double *A,
                        double *G, double *q, double *b, double *g, double *d, double *u,
                        int nthreads) {
    if (!cfg || nprob < 0) return SSQP_ERR_ARG;
    const size_t N = cfg->N, M = cfg->M, J = cfg->J;
    std::atomic<int> rc{SSQP_OK};
    parallel_for(nprob, nthreads, [&](int p) {
        const size_t P = p;
        int r = ssqp_generate_problem(cfg, seed0 + P, V ? V + P * N * N : nullptr, A ? A + P * M * N : nullptr,
                                      G ? G + P * J * N : nullptr, q + P * N, b ? b + P * M : nullptr,
                                      g ? g + P * J : nullptr, d + P * N, u + P * N);
        if (r != SSQP_OK) rc = r;
    });
    return rc;
}

int ssqp_phase1_f64(int N, int M, int J, const double *A, const double *G, const double *b,
                    const double *g, const double *d, const double *u,
                    const ssqp_settings *settingsLP, double *x0, int32_t *S, int32_t *status) {
    if (N <= 0 || M < 0 || J < 0 || !d || !u || !x0 || !S || !status) return SSQP_ERR_ARG;
    if ((M > 0 && (!A || !b)) || (J > 0 && (!G || !g))) return SSQP_ERR_ARG;
    ssqp_settings def;
    ssqp_default_settings(&def);
    const ssqp_settings *st = settingsLP ? settingsLP : &def;
    if (st->rule != 0) return SSQP_ERR_UNSUPPORTED;
    *status = phase1_one(N, M, J, A, G, b, g, d, u, st->tol, x0, S);
    return SSQP_OK;
}

int ssqp_phase1_batch_f64(int nprob, int N, int M, int J, const double *A, const double *G,
                          const double *b, const double *g, const double *d, const double *u,
                          const ssqp_settings *settingsLP, double *x0, int32_t *S, int32_t *status,
                          int nthreads) {
    if (nprob < 0) return SSQP_ERR_ARG;
    std::atomic<int> rc{SSQP_OK};
    const size_t n = N, m = M, j = J;
    parallel_for(nprob, nthreads, [&](int p) {
        const size_t P = p;
        int r = ssqp_phase1_f64(N, M, J, A ? A + P * m * n : nullptr, G ? G + P * j * n : nullptr,
                                b ? b + P * m : nullptr, g ? g + P * j : nullptr, d + P * n, u + P * n,
                                settingsLP, x0 + P * n, S + P * (n + j), status + p);
        if (r != SSQP_OK) rc = r;
    });
    return rc;
}

}  // extern "C"
