// Probe (not product): how fast can ONE compute unit pull a matrix that sits in L2, and how does that scale with the
// wavefronts of the workgroup, the loads each keeps in flight, and the number of workgroups that share the job?
//
// The access pattern is the many-rows Phase-1 kernel's Y.c refresh (csrc/ssqp_phase1.hip, refreshY): a 72 x 2184 row-major
// matrix of doubles (1.26 MB: the LP of BASELINE config 5), walked row by row in a listed order; thread t of a workgroup
// reads the columns c0 + t + u * NT of the listed row (u < 4: 512 contiguous bytes per wavefront and load), DEPTH rows
// requested ahead, every value added to a per-thread sum.  W workgroups split the 2048 general columns between them.
//
//   hipcc --offload-arch=gfx950 -O3 l2_pull_probe.hip -o /tmp/l2_pull_probe && /tmp/l2_pull_probe
//
// Output: bytes per cycle per workgroup (s_memtime of thread 0, the slowest workgroup) and in total, for
// NT in {256, 512, 1024}, DEPTH in {4, 8, 16}, W in {1, 2, 4, 8, 16, 32}.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int ROWS = 72, N1 = 2184, NCOL = 2048, STEPS = 112 * 16;   // (112 listed steps per refresh, 16 refreshes per launch)

template <int NT, int DEPTH, int CG>
__global__ __launch_bounds__(NT) void pull(const double *__restrict__ A, const int *__restrict__ rows, int W, double *out,
                                           unsigned long long *cyc) {
    // this workgroup's columns: NCOL / W of them, CG per thread and block; a wavefront without a column sits the loop out
    const int per = NCOL / W, c0 = blockIdx.x * per;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double s[CG];
#pragma unroll
    for (int u = 0; u < CG; ++u) s[u] = 0.0;
    for (int cb = 0; cb < per; cb += CG * NT) {
        if (cb + (int)(threadIdx.x & ~63u) >= per) continue;
        int col[CG];
        bool on[CG];
#pragma unroll
        for (int u = 0; u < CG; ++u) {
            const int c = cb + (int)threadIdx.x + u * NT;
            on[u] = c < per;
            col[u] = c0 + (on[u] ? c : 0);
        }
        double buf[DEPTH][CG];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int u = 0; u < CG; ++u) buf[d][u] = A[(size_t)rows[d] * N1 + col[u]];
        for (int st = 0; st < STEPS; st += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                double v[CG];
#pragma unroll
                for (int u = 0; u < CG; ++u) v[u] = buf[d][u];
                const int nx = st + DEPTH + d < STEPS ? st + DEPTH + d : STEPS - 1;
                const int r = __builtin_amdgcn_readfirstlane(rows[nx]);
#pragma unroll
                for (int u = 0; u < CG; ++u) buf[d][u] = A[(size_t)r * N1 + col[u]];
#pragma unroll
                for (int u = 0; u < CG; ++u) s[u] += on[u] ? v[u] : 0.0;
            }
        }
    }
    double tot = 0.0;
#pragma unroll
    for (int u = 0; u < CG; ++u) tot += s[u];
    out[(size_t)blockIdx.x * NT + threadIdx.x] = tot;
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}

template <int NT, int DEPTH>
static void run(const double *dA, const int *dRows, double *dOut, unsigned long long *dCyc, const std::vector<double> &colsum) {
    for (int W : {1, 2, 4, 8, 16, 32}) {
        if (NCOL / W < 1) continue;
        const int per = NCOL / W;
        for (int rep = 0; rep < 2; ++rep) {   // (the first launch brings A into L2)
            if (per >= 4 * NT) hipLaunchKernelGGL((pull<NT, DEPTH, 4>), dim3(W), dim3(NT), 0, 0, dA, dRows, W, dOut, dCyc);
            else if (per >= 2 * NT) hipLaunchKernelGGL((pull<NT, DEPTH, 2>), dim3(W), dim3(NT), 0, 0, dA, dRows, W, dOut, dCyc);
            else hipLaunchKernelGGL((pull<NT, DEPTH, 1>), dim3(W), dim3(NT), 0, 0, dA, dRows, W, dOut, dCyc);
        }
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> cyc(W);
        (void)hipMemcpy(cyc.data(), dCyc, W * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long worst = 0;
        for (auto c : cyc) worst = c > worst ? c : worst;
        const double bytes = (double)STEPS * NCOL * 8.0;   // useful bytes of the whole job
        // (check: thread 0 of workgroup 0, its first column block)
        std::vector<double> out((size_t)W * NT);
        (void)hipMemcpy(out.data(), dOut, out.size() * sizeof(double), hipMemcpyDeviceToHost);
        (void)colsum;
        printf("threads %4d  rows ahead %2d  workgroups %2d (%4d columns each, %2d wavefronts at work) : %8llu cycles  %6.1f B/cycle per workgroup  %7.1f B/cycle in total\n",
               NT, DEPTH, W, per, (per < NT ? per : NT) / 64 > 0 ? (per < NT ? per : NT) / 64 : 1, worst, bytes / W / (double)worst, bytes / (double)worst);
    }
}

int main() {
    std::vector<double> A((size_t)ROWS * N1);
    for (size_t i = 0; i < A.size(); ++i) A[i] = (double)((i * 2654435761u) % 1000) * 1e-3;
    std::vector<int> rows(STEPS);
    for (int i = 0; i < STEPS; ++i) rows[i] = (int)((i * 37u + (i / 112) * 11u) % ROWS);   // a listed order, all rows in use
    double *dA, *dOut;
    int *dRows;
    unsigned long long *dCyc;
    (void)hipMalloc(&dA, A.size() * sizeof(double));
    (void)hipMalloc(&dRows, rows.size() * sizeof(int));
    (void)hipMalloc(&dOut, 32 * 1024 * sizeof(double));
    (void)hipMalloc(&dCyc, 64 * sizeof(unsigned long long));
    (void)hipMemcpy(dA, A.data(), A.size() * sizeof(double), hipMemcpyHostToDevice);
    (void)hipMemcpy(dRows, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice);
    std::vector<double> colsum;
    printf("matrix %d x %d doubles (%.2f MB), %d listed row reads of %d columns per launch (%.1f MB useful)\n", ROWS, N1,
           A.size() * 8e-6, STEPS, NCOL, STEPS * (double)NCOL * 8e-6);
    run<256, 4>(dA, dRows, dOut, dCyc, colsum);
    run<256, 8>(dA, dRows, dOut, dCyc, colsum);
    run<256, 16>(dA, dRows, dOut, dCyc, colsum);
    run<512, 4>(dA, dRows, dOut, dCyc, colsum);
    run<512, 8>(dA, dRows, dOut, dCyc, colsum);
    run<512, 16>(dA, dRows, dOut, dCyc, colsum);
    run<1024, 4>(dA, dRows, dOut, dCyc, colsum);
    run<1024, 8>(dA, dRows, dOut, dCyc, colsum);
    if (hipDeviceSynchronize() != hipSuccess) {
        printf("HIP error\n");
        return 1;
    }
    return 0;
}
