// Probe (not product): what does one publish -> collect exchange between a main workgroup and H helper workgroups of the SAME
// launch cost on MI355X -- the exchange DESIGN.md section 7 (next, 3) sketches for one large QP spread over several CUs?
//
//   main (block 0):   writes a payload of PB bytes to global memory, then epoch = it (release, agent scope);
//                     waits until every helper's ack word equals it (acquire loads) -- or, COUNTER mode, until ONE counter
//                     the helpers add to has reached it * H;
//   helper (block h): waits until epoch == it (acquire), reads the payload, checks it, ack[h] = it (release) / done += 1.
//
// Every wait is BOUNDED (a workgroup that is not resident, or a bug, ends in an error code, never in a hang), the
// payload check verifies that a release / acquire pair at agent scope makes plain stores visible across XCDs.
//
//   hipcc --offload-arch=gfx950 -O3 wg_exchange_probe.hip -o /tmp/wg_exchange_probe && /tmp/wg_exchange_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NT = 256, ITER = 200;
constexpr long SPIN_MAX = 4000000;   // (~ a second of polling at worst)

struct Ctl {
    unsigned epoch;
    unsigned err;
    unsigned pad[30];
    unsigned done;
    unsigned pad2[31];
    unsigned ack[64 * 32];   // one 128-byte line per helper
};

template <bool COUNTER>
__global__ __launch_bounds__(NT) void exchange(Ctl *ctl, double *payload, int nPay, unsigned long long *cycles) {
    const int tid = threadIdx.x, H = gridDim.x - 1;
    __shared__ int ok;
    if (blockIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 1; it <= ITER; ++it) {
            for (int i = tid; i < nPay; i += NT) payload[i] = (double)(it * 7 + i);
            __syncthreads();   // (every thread's stores are issued; the release below makes them visible)
            if (tid == 0) {
                __hip_atomic_store(&ctl->epoch, (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                int good = 1;
                if (COUNTER) {
                    long spin = 0;
                    while (__hip_atomic_load(&ctl->done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)(it * H)) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spin > SPIN_MAX) {
                            good = 0;
                            break;
                        }
                    }
                } else
                for (int h = 0; h < H && good; ++h) {
                    long spin = 0;
                    while (__hip_atomic_load(&ctl->ack[h * 32], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)it) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spin > SPIN_MAX) {
                            good = 0;
                            break;
                        }
                    }
                }
                ok = good;
                if (!good) __hip_atomic_store(&ctl->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            if (!ok) break;
        }
        if (tid == 0) {
            cycles[0] = __builtin_amdgcn_s_memtime() - t0;
            __hip_atomic_store(&ctl->epoch, 0xffffffffu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // (exit)
        }
    } else {
        const int h = blockIdx.x - 1;
        for (int it = 1; it <= ITER; ++it) {
            if (tid == 0) {
                long spin = 0;
                int good = 1;
                for (;;) {
                    const unsigned e = __hip_atomic_load(&ctl->epoch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    if (e == 0xffffffffu) {
                        good = 0;
                        break;
                    }
                    if (e >= (unsigned)it) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spin > SPIN_MAX) {
                        good = 0;
                        break;
                    }
                }
                ok = good;
            }
            __syncthreads();
            if (!ok) return;
            // (the acquire above was thread 0's; the barrier orders the other threads behind it in this workgroup -- and a
            //  fence makes their loads bypass stale lines)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            int bad = 0;
            for (int i = tid; i < nPay; i += NT) {
                const double v = __builtin_nontemporal_load(payload + i);
                // (the main workgroup may already be writing the NEXT payload only after this helper's ack: exact check)
                if (v != (double)(it * 7 + i)) bad = 1;
            }
            if (bad) __hip_atomic_store(&ctl->err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (tid == 0) {
                if (COUNTER) (void)__hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(&ctl->ack[h * 32], (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

int main() {
    Ctl *ctl;
    double *payload;
    unsigned long long *cyc;
    (void)hipMalloc(&ctl, sizeof(Ctl));
    (void)hipMalloc(&payload, 64 * 1024);
    (void)hipMalloc(&cyc, 8);
    for (int counter = 0; counter < 2; ++counter)
    for (int H : {1, 4, 8, 16, 31}) {
        for (int bytes : {0, 8 * 1024, 48 * 1024}) {
            (void)hipMemset(ctl, 0, sizeof(Ctl));
            if (counter) hipLaunchKernelGGL(exchange<true>, dim3(1 + H), dim3(NT), 0, 0, ctl, payload, bytes / 8, cyc);
            else hipLaunchKernelGGL(exchange<false>, dim3(1 + H), dim3(NT), 0, 0, ctl, payload, bytes / 8, cyc);
            if (hipDeviceSynchronize() != hipSuccess) {
                printf("HIP error\n");
                return 1;
            }
            unsigned long long c = 0;
            Ctl host;
            (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&host, ctl, sizeof(Ctl), hipMemcpyDeviceToHost);
            printf("%s  helpers %2d  payload %5d B : %7.0f cycles per publish -> all done  (err %u: 0 = every payload read back exact)\n",
                   counter ? "one counter  " : "ack per helper", H, bytes, (double)c / ITER, host.err);
        }
    }
    return 0;
}
