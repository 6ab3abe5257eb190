// gather_probe.hip -- what rate can a row gather reach when every source row is L2-resident?  Each lane group of LPR
// lanes reads U random rows (LPR*16 B each) of a per-XCD window per iteration; indices come from an LCG so no index
// stream competes.  MODE 0: global_load_dwordx4 into registers.  MODE 1: global_load_lds_dwordx4 (LDS-DMA) into a
// per-wave staging area, read back with ds_read_b128.  MODE 2: streaming (consecutive rows) for the L2 ceiling.
//   hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip && ./gather_probe
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

__device__ __forceinline__ float4 add4(float4 x, float4 a) {
    a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
    return a;
}

template <int LPR, int U, int MODE>
__global__ __launch_bounds__(256) void gather_probe(const float4 *__restrict__ x, uint32_t win_rows, uint32_t stride4,
                                                    int iters, float4 *__restrict__ out) {
    __shared__ float4 stage[MODE == 1 ? 4 * U * 64 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane % LPR;
    const uint32_t gid = (blockIdx.x * 256 + threadIdx.x) / LPR;
    const uint32_t xcd = blockIdx.x & 7u;
    const float4 *xw = x + (size_t)xcd * win_rows * stride4;
    uint32_t state = gid * 2654435761u + 12345u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0 || MODE == 2) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t r;
                if constexpr (MODE == 0) {
                    state = state * 1664525u + 1013904223u;
                    r = (uint32_t)(((uint64_t)(state >> 4) * win_rows) >> 28);
                } else {
                    r = (gid + (uint32_t)(it * U + u) * 1031u) % win_rows;
                }
                v[u] = xw[(size_t)r * stride4 + li];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc = add4(v[u], acc);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                state = state * 1664525u + 1013904223u;
                const uint32_t r = (uint32_t)(((uint64_t)(state >> 4) * win_rows) >> 28);
                const float4 *src = xw + (size_t)r * stride4 + li;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(stage + (wave * U + u) * 64),
                                                 16, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
#pragma unroll
            for (int u = 0; u < U; ++u) acc = add4(stage[(wave * U + u) * 64 + lane], acc);
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int LPR, int U, int MODE>
static void run(const float4 *dx, uint32_t win_rows, uint32_t stride4, float4 *dout, int wg_per_cu) {
    const int nwg = 256 * wg_per_cu, iters = 2000 / U;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((gather_probe<LPR, U, MODE>), dim3(nwg), dim3(256), 0, 0, dx, win_rows, stride4, iters, dout);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)nwg * 256 * iters * U * 16;
    printf("mode=%d lanes/row=%2d (%4d B) U=%2d window=%5.2f MB stride=%4u B wg/cu=%d: %7.3f ms  %6.2f TB/s\n", MODE, LPR,
           LPR * 16, U, (double)win_rows * LPR * 16 / 1048576.0, stride4 * 16, wg_per_cu, ms, bytes / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

int main() {
    const uint32_t stride4 = 152;                 // 608 floats per row, as x at F=602
    const size_t rows_total = 8u * 65536u;
    float4 *dx, *dout;
    CK(hipMalloc(&dx, rows_total * stride4 * 16));
    CK(hipMemset(dx, 0, rows_total * stride4 * 16));
    CK(hipMalloc(&dout, (size_t)256 * 8 * 256 * 16));
    // window sizes in rows chosen for ~2 MB of touched bytes
    printf("-- registers, 2 MB window, lanes/row x in-flight\n");
    run<8, 4, 0>(dx, 16384, stride4, dout, 8);
    run<8, 8, 0>(dx, 16384, stride4, dout, 8);
    run<16, 4, 0>(dx, 8192, stride4, dout, 8);
    run<16, 8, 0>(dx, 8192, stride4, dout, 8);
    run<32, 4, 0>(dx, 4096, stride4, dout, 8);
    run<32, 8, 0>(dx, 4096, stride4, dout, 8);
    run<32, 16, 0>(dx, 4096, stride4, dout, 8);
    run<64, 4, 0>(dx, 2048, stride4, dout, 8);
    run<64, 8, 0>(dx, 2048, stride4, dout, 8);
    printf("-- window size (512-B rows, U=8)\n");
    run<32, 8, 0>(dx, 1024, stride4, dout, 8);
    run<32, 8, 0>(dx, 2048, stride4, dout, 8);
    run<32, 8, 0>(dx, 6144, stride4, dout, 8);
    run<32, 8, 0>(dx, 8192, stride4, dout, 8);
    run<32, 8, 0>(dx, 10240, stride4, dout, 8);
    run<32, 8, 0>(dx, 16384, stride4, dout, 8);
    printf("-- occupancy (512-B rows, 2 MB)\n");
    run<32, 8, 0>(dx, 4096, stride4, dout, 2);
    run<32, 8, 0>(dx, 4096, stride4, dout, 4);
    printf("-- dense rows (stride = row bytes)\n");
    run<32, 8, 0>(dx, 4096, 32, dout, 8);
    run<16, 8, 0>(dx, 8192, 16, dout, 8);
    printf("-- LDS-DMA gather\n");
    run<16, 4, 1>(dx, 8192, stride4, dout, 8);
    run<32, 4, 1>(dx, 4096, stride4, dout, 8);
    run<32, 8, 1>(dx, 4096, stride4, dout, 8);
    run<64, 8, 1>(dx, 2048, stride4, dout, 8);
    printf("-- streaming from the window (L2 ceiling)\n");
    run<64, 8, 2>(dx, 2048, 64, dout, 8);
    run<32, 8, 2>(dx, 4096, 32, dout, 8);
    return 0;
}
