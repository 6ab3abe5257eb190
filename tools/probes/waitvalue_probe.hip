// waitvalue_probe.hip -- do hipStreamWaitValue32 / hipStreamWriteValue32 order two streams of one process on device-side
// flags, with the WAIT enqueued before the write (no host synchronisation, no event recorded yet)?  That is what the
// in-process device transport (dory_comm_init_local, round 6) rests on.  Prints one verdict per flag-memory kind.
//   hipcc --offload-arch=gfx950 -O3 -o waitvalue_probe waitvalue_probe.hip && timeout 60 ./waitvalue_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);       \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

__global__ void spin_then_store(uint32_t *p, uint32_t v, long long cycles) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(10);
    *p = v;
}
__global__ void copy_word(const uint32_t *src, uint32_t *dst) { *dst = *src; }

static int trial(uint32_t *flag, const char *kind) {
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    uint32_t *data, *seen;
    CK(hipMalloc(&data, 4)); CK(hipMalloc(&seen, 4));
    CK(hipMemset(data, 0, 4)); CK(hipMemset(seen, 0, 4)); CK(hipMemset(flag, 0, 4));
    CK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    // consumer first: waits for the flag, then reads what the producer wrote before raising it
    hipError_t e = hipStreamWaitValue32(a, flag, 1, hipStreamWaitValueGte, 0xFFFFFFFFu);
    if (e != hipSuccess) { printf("%-28s hipStreamWaitValue32 refused: %s\n", kind, hipGetErrorString(e)); return 1; }
    hipLaunchKernelGGL(copy_word, dim3(1), dim3(1), 0, a, data, seen);
    const double enq_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // producer: ~5 ms of work (100 MHz wall clock), the data, then the flag
    hipLaunchKernelGGL(spin_then_store, dim3(1), dim3(1), 0, b, data, 42u, 500000LL);
    e = hipStreamWriteValue32(b, flag, 1, 0);
    if (e != hipSuccess) { printf("%-28s hipStreamWriteValue32 refused: %s\n", kind, hipGetErrorString(e)); return 1; }
    CK(hipStreamSynchronize(a));
    const double tot_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    uint32_t h = 0;
    CK(hipMemcpy(&h, seen, 4, hipMemcpyDeviceToHost));
    printf("%-28s consumer saw %u (want 42); enqueue of wait+kernel returned after %.3f ms, consumer done after %.2f ms (producer ~5 ms)  %s\n",
           kind, h, enq_ms, tot_ms, h == 42 && enq_ms < 2.0 && tot_ms > 3.0 ? "OK" : "UNEXPECTED");
    // many rounds, both directions (a ping-pong of 200 sequence numbers), as the transport does per exchange
    CK(hipMemset(flag, 0, 8));
    CK(hipDeviceSynchronize());
    const auto t1 = std::chrono::steady_clock::now();
    for (uint32_t s = 1; s <= 200; ++s) {
        CK(hipStreamWaitValue32(a, flag + 1, s - 1, hipStreamWaitValueGte, 0xFFFFFFFFu));
        CK(hipStreamWriteValue32(a, flag, s, 0));
        CK(hipStreamWaitValue32(b, flag, s, hipStreamWaitValueGte, 0xFFFFFFFFu));
        CK(hipStreamWriteValue32(b, flag + 1, s, 0));
    }
    CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
    const double pp_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    uint32_t hf[2];
    CK(hipMemcpy(hf, flag, 8, hipMemcpyDeviceToHost));
    printf("%-28s 200 ping-pongs: flags %u %u, %.1f us per round trip\n", kind, hf[0], hf[1], pp_ms * 1e3 / 200);
    CK(hipFree(data)); CK(hipFree(seen));
    CK(hipStreamDestroy(a)); CK(hipStreamDestroy(b));
    return h == 42 ? 0 : 1;
}

int main() {
    int can = -1;
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    int bad = 0;
    uint32_t *f1 = nullptr, *f2 = nullptr, *f3 = nullptr;
    CK(hipMalloc(&f1, 256));
    bad += trial(f1, "hipMalloc");
    if (hipExtMallocWithFlags((void **)&f2, 8, hipMallocSignalMemory) == hipSuccess) bad += trial(f2, "hipMallocSignalMemory (8 B)");
    else printf("hipMallocSignalMemory: allocation refused\n");
    if (hipHostMalloc((void **)&f3, 256, hipHostMallocMapped) == hipSuccess) bad += trial(f3, "hipHostMalloc (mapped)");
    printf("verdict: %s\n", bad ? "SOME KIND FAILED" : "all kinds ordered the streams");
    return 0;
}
