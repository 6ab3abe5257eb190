// mfma_peak_probe.hip -- what does v_mfma_f32_32x32x2_f32 sustain on this chip with nothing else in the way?  K2's roofline
// is priced against 157 TFLOP/s (256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz); an MFMA-only loop at K2's occupancy shows
// the clock the matrix pipes really hold under load, i.e. the ceiling a GEMM can be asked to approach.
//   hipcc --offload-arch=gfx950 -O3 mfma_peak_probe.hip -o mfma_peak_probe && ./mfma_peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;   // never true: keeps the loop alive
}

template <int NACC>
static void run(int wgs_per_cu, int iters) {
    float *d;
    hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, d, iters / 8, 1.f, 1.f);   // warm-up
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 1.f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flops = (double)grid * 4 /*waves*/ * iters * NACC * (2.0 * 32 * 32 * 2);
    printf("%d workgroups of 4 waves per CU, %d independent accumulators per wave, %d iterations: %.3f ms  %.1f TFLOP/s  (%.3f of 157.3; implied clock at 64 FLOP/clk/SIMD: %.2f GHz)\n",
           wgs_per_cu, NACC, iters, best, flops / best / 1e9, flops / best / 1e9 / 157.3, flops / best / 1e9 / 157.3 * 2.4);
    hipFree(d);
}

int main() {
    run<4>(1, 20000);
    run<4>(2, 20000);
    run<4>(4, 20000);    // K2's occupancy (gemm_kernel_occ4) and accumulator count (2 x 2 tiles per wave)
    run<2>(4, 20000);
    run<1>(4, 20000);
    run<4>(4, 200000);   // ~60 ms of MFMA: does the clock sag under sustained load?
    return 0;
}
