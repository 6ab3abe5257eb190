// sync_probe.hip -- cost per step of the per-XCD "start step b when everybody finished step b-SLACK-1" protocol with
// no work in the steps: G workgroups per XCD (workgroup i -> XCD i & 7), one arrival atomic per workgroup and step
// (L2-local), one poller lane per workgroup.  Variants: poll sleep, arrival scope, threads per workgroup.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);       \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

template <int SLEEP, int MODE>
__global__ void sync_probe(uint32_t *done, uint32_t G, int steps, int slack, uint32_t *timeouts, int work) {
    const uint32_t xcd = blockIdx.x & 7u;
    uint32_t *dq = done + (size_t)xcd * steps * (MODE == 2 ? 32 : 1);
    volatile float sink = 0.f;
    for (int b = 0; b < steps; ++b) {
        if (b > slack && threadIdx.x == 0) {
            int spins = 0;
            if constexpr (MODE == 2) {   // arrivals spread over 32 words (one 128-B line), the poller adds them up
                const uint32_t *w = dq + (size_t)(b - slack - 1) * 32;
                while (true) {
                    uint32_t sum = 0;
                    for (int i = 0; i < 32; i += 4) {
                        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                        u4 v;
                        const uint32_t *p = w + i;
                        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
                        sum += v.x + v.y + v.z + v.w;
                    }
                    if (sum >= G) break;
                    __builtin_amdgcn_s_sleep(SLEEP);
                    if (++spins > 100000) { atomicAdd(timeouts, 1u); break; }
                }
            } else {
                const uint32_t *w = dq + (b - slack - 1);
                while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < G) {
                    __builtin_amdgcn_s_sleep(SLEEP);
                    if (++spins > 100000) { atomicAdd(timeouts, 1u); break; }
                }
            }
        }
        __syncthreads();
        for (int i = 0; i < work; ++i) sink = sink + 1.f;
        __syncthreads();
        if (threadIdx.x == 0) {
            if constexpr (MODE == 0) __hip_atomic_fetch_add(dq + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if constexpr (MODE == 1) __hip_atomic_fetch_add(dq + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(dq + (size_t)b * 32 + ((blockIdx.x >> 3) & 31), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

template <int SLEEP, int MODE>
static void run(uint32_t *done, uint32_t *d_to, uint32_t G, int threads, int slack, int work) {
    const int steps = 400;
    CK(hipMemset(done, 0, (size_t)8 * steps * 32 * 4));
    CK(hipMemset(d_to, 0, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((sync_probe<SLEEP, MODE>), dim3(8 * G), dim3(threads), 0, 0, done, G, steps, slack, d_to, work);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; uint32_t to;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&to, d_to, 4, hipMemcpyDeviceToHost));
    printf("mode %d sleep %3d G=%3u threads %4d slack %d work %4d: %7.2f us/step  timeouts %u\n", MODE, SLEEP, G, threads, slack,
           work, ms * 1e3 / steps, to);
    fflush(stdout);
}

int main() {
    uint32_t *done, *d_to;
    CK(hipMalloc(&done, (size_t)8 * 400 * 32 * 4));
    CK(hipMalloc(&d_to, 4));
    for (uint32_t G : {32u, 64u, 128u, 160u}) {
        run<32, 0>(done, d_to, G, 256, 1, 0);
        run<4, 0>(done, d_to, G, 256, 1, 0);
        run<1, 0>(done, d_to, G, 256, 1, 0);
        run<4, 1>(done, d_to, G, 256, 1, 0);
        run<4, 2>(done, d_to, G, 256, 1, 0);
        run<4, 0>(done, d_to, G, 256, 0, 0);
        run<4, 0>(done, d_to, G, 256, 2, 0);
    }
    run<4, 0>(done, d_to, 32, 1024, 1, 0);
    run<4, 2>(done, d_to, 32, 1024, 1, 0);
    run<4, 0>(done, d_to, 160, 256, 1, 2000);
    run<4, 2>(done, d_to, 160, 256, 1, 2000);
    return 0;
}
