// ssqp_phase1_wave.hip -- gfx950: the stand-alone Phase-1 kernel with one 64-lane wavefront per QP (ssqp_phase1_wave.h has
// the algorithm; reference: initQP src/SSQP.jl:461-560 + cDantzigLP src/Simplex.jl:445-615).  One wavefront per SIMD
// (every column of the LP lives in registers: up to ~400 per lane), one QP per 64-thread block -- the dispatcher hands
// a SIMD its next QP as soon as one finishes, so uneven pivot counts balance themselves.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssqp_hip.h"
#include "ssqp_internal.h"
#include "ssqp_device.h"
#include "ssqp_phase1_wave.h"

namespace ssqp {
namespace {

template <int NC, int MC>
__global__ __launch_bounds__(64, 1) void ssqp_phase1_wave_kernel(p1w::Params P) {
    __shared__ __attribute__((aligned(16))) double lds[p1w::lds_doubles<NC, MC>()];
    const int prob = blockIdx.x;
    if (prob >= P.nprob) return;
    int st;
    (void)p1w::solve_one<NC, MC>(P, prob, lds, st);
}

template <int NC, int MC>
hipError_t launch(const p1w::Params &P, hipStream_t stream) {
    hipLaunchKernelGGL((ssqp_phase1_wave_kernel<NC, MC>), dim3(P.nprob), dim3(64), 0, stream, P);
    return hipGetLastError();
}

}  // namespace

bool phase1_wave_applies(int N, int M, int J) {
    const int M0 = M + J;
    return M0 >= 1 && M0 <= 11 && N + J + M0 <= 64 * 9;
}

hipError_t launch_phase1_wave(int nprob, int N, int M, int J, const double *A, const double *G, const double *b, const double *g,
                              const double *d, const double *u, double tol, double *x0, int32_t *S, int32_t *status,
                              unsigned int *fbCount, int *fbList, hipStream_t stream) {
    p1w::Params P;
    P.nprob = nprob; P.N = N; P.M = M; P.J = J;
    P.A = A; P.G = G; P.b = b; P.g = g; P.d = d; P.u = u;
    P.tol = tol;
    P.x0 = x0; P.S = S; P.status = status;
    P.fbCount = fbCount; P.fbList = fbList;
    const int M0 = M + J, N1 = N + J + M0;
    // builds by column slots (64 columns each) and rows: the loops over both are unrolled over register arrays
    if (N1 <= 64 * 5) return M0 <= 4 ? launch<5, 4>(P, stream) : (M0 <= 8 ? launch<5, 8>(P, stream) : launch<5, 11>(P, stream));
    return M0 <= 4 ? launch<9, 4>(P, stream) : (M0 <= 8 ? launch<9, 8>(P, stream) : launch<9, 11>(P, stream));
}

#ifdef SSQP_PHASE_PROFILE
int phase1_wave_debug_phases(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(p1w::g_p1wphase), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        static unsigned long long zero[16];
        if (hipMemcpyToSymbol(HIP_SYMBOL(p1w::g_p1wphase), zero, sizeof(zero)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

}  // namespace ssqp

#ifdef SSQP_PHASE_PROFILE
extern "C" int ssqp_debug_phase1_wave_phases(unsigned long long *out16, int reset) { return ssqp::phase1_wave_debug_phases(out16, reset); }
#endif
