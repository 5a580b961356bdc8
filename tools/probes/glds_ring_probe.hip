// Probe (not product): one wavefront streams columns of doubles from global memory through an LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4, inline asm, counted s_waitcnt vmcnt) and sums them with weights -- correctness against the
// host and the bandwidth 1024 such wavefronts reach together.   hipcc --offload-arch=gfx950 -O3 glds_ring_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NI = 2;      // 1 KiB pieces per ring slot (2 KiB = one column of 256 doubles)
constexpr int D = 8;       // ring slots

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// issue column `col` (len doubles, len <= 128 * NI) into ring slot `slot`
__device__ __forceinline__ void issue(const double *col, int len, unsigned ringBase, int slot) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = 128 * i + 2 * lane;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ringBase + (unsigned)(slot * NI + i) * 1024u);
        if (e < len) glds16(col + e, dst);
        else asm volatile("s_nop 0" ::: "memory");
    }
}

__global__ __launch_bounds__(64, 1) void probe(const double *__restrict__ base, const int *__restrict__ colidx, const double *__restrict__ w,
                                               int ncols, int len, size_t stride, double *out, int nrep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const unsigned ringBase = (unsigned)(size_t)smem;
    const double *ring = reinterpret_cast<const double *>(smem);
    const double *mine = base + (size_t)blockIdx.x * stride;
    const int *ci = colidx + (size_t)blockIdx.x * ncols;
    double2 acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = make_double2(0.0, 0.0);
    for (int rep = 0; rep < nrep; ++rep) {
        // prologue
        for (int c = 0; c < D && c < ncols; ++c) issue(mine + (size_t)ci[c] * len, len, ringBase, c);
        for (int c = 0; c < ncols; ++c) {
            if (c + D - 1 < ncols) wait_vm<NI * (D - 1)>();   // column c has landed; D - 1 younger ones may be in flight
            else wait_vm<0>();
            const int slot = c % D;
            const double wc = w[c];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int e = 128 * i + 2 * lane;
                if (e < len) {
                    const double2 v = *reinterpret_cast<const double2 *>(ring + (slot * NI + i) * 128 + 2 * lane);
                    acc[i].x = fma(v.x, wc, acc[i].x);
                    acc[i].y = fma(v.y, wc, acc[i].y);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's reads are done before it is refilled
            if (c + D < ncols) issue(mine + (size_t)ci[c + D] * len, len, ringBase, slot);
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int e = 128 * i + 2 * lane;
        if (e < len) {
            out[(size_t)blockIdx.x * len + e] = acc[i].x;
            out[(size_t)blockIdx.x * len + e + 1] = acc[i].y;
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    const int nwave = 1024, len = 256, ntot = 256, ncols = 200, nrep = 20;
    const size_t stride = (size_t)ntot * len;
    std::vector<double> h(stride * nwave), w(ncols);
    std::vector<int> ci((size_t)nwave * ncols);
    srand(1);
    for (auto &x : h) x = (rand() % 1000) / 1000.0;
    for (auto &x : w) x = (rand() % 100) / 100.0;
    for (int b = 0; b < nwave; ++b)
        for (int c = 0; c < ncols; ++c) ci[(size_t)b * ncols + c] = (c * 7 + b) % ntot;
    double *d, *dw, *dout;
    int *dci;
    CK(hipMalloc(&d, h.size() * 8)); CK(hipMalloc(&dw, ncols * 8)); CK(hipMalloc(&dout, (size_t)nwave * len * 8));
    CK(hipMalloc(&dci, ci.size() * 4));
    CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, w.data(), ncols * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dci, ci.data(), ci.size() * 4, hipMemcpyHostToDevice));
    const size_t lds = 40 * 1024;   // (40 KiB per wavefront: four per CU, as in the solver)
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe, dim3(nwave), dim3(64), lds, 0, d, dci, dw, ncols, len, stride, dout, nrep);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)nwave * ncols * len * 8 * nrep;
        printf("run %d: %.3f ms, %.2f TB/s (%d wavefronts, %d columns of %d B, ring %d x %d KiB)\n", it, ms, bytes / ms / 1e9, nwave,
               ncols, len * 8, D, NI);
    }
    std::vector<double> o((size_t)nwave * len);
    CK(hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int b = 0; b < nwave; b += 37)
        for (int e = 0; e < len; ++e) {
            double s = 0;
            for (int c = 0; c < ncols; ++c) s = fma(h[(size_t)b * stride + (size_t)ci[(size_t)b * ncols + c] * len + e], w[c], s);
            s *= nrep;
            double ref = 0;  // (the kernel accumulates rep after rep: same value up to rounding)
            (void)ref;
            const double err = fabs(o[(size_t)b * len + e] - s) / fmax(1.0, fabs(s));
            if (err > maxerr) maxerr = err;
        }
    printf("max rel err vs host: %.3e  %s\n", maxerr, maxerr < 1e-9 ? "OK" : "MISMATCH");
    return maxerr < 1e-9 ? 0 : 2;
}
