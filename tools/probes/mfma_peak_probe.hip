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

// the same loop with the accumulators pinned in ARCH VGPRs (inline asm, "+v") or in AccVGPRs ("+a"): K2's kernels keep
// theirs in VGPRs (NumAgprs: 0) and plateau at 63 % MFMA-busy; does the register class matter?
template <int NACC, bool AGPR>
__global__ __launch_bounds__(256) void mfma_loop_cls(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NACC, bool AGPR>
static void run_cls(int wgs_per_cu, int iters) {
    float *d;
    hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((mfma_loop_cls<NACC, AGPR>), dim3(grid), dim3(256), 0, 0, d, iters / 8, 1.f, 1.f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mfma_loop_cls<NACC, AGPR>), dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 1.f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flops = (double)grid * 4 * iters * NACC * (2.0 * 32 * 32 * 2);
    printf("accumulators in %s, %d workgroups of 4 waves per CU, %d accumulators per wave: %.3f ms  %.1f TFLOP/s  (%.3f of 157.3)\n",
           AGPR ? "AccVGPRs" : "arch VGPRs", wgs_per_cu, NACC, best, flops / best / 1e9, flops / best / 1e9 / 157.3);
    hipFree(d);
}

// operands as the GEMM kernels get them: the B fragment of every pair of MFMAs comes out of LDS (ds_read2_b32 -> wait -> two MFMAs:
// the schedule hipcc emits for K2), AHEAD = 0; or software-pipelined by hand, the next pair's fragment requested before this pair's
// MFMAs issue (AHEAD = 1)
template <int AHEAD>
__global__ __launch_bounds__(256) void mfma_lds_loop(float *out, int iters, float a0) {
    __shared__ float bs[64 * 68];
    for (int i = threadIdx.x; i < 64 * 68; i += 256) bs[i] = 1.f + i * 1e-6f;
    __syncthreads();
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int lane = threadIdx.x & 63;
    const float *bp = bs + (lane >> 5) * 68 + (lane & 31);
    float a = a0 + threadIdx.x * 1e-6f;
    float b0 = bp[0], b1 = bp[32];
    for (int it = 0; it < iters; ++it) {
        const int k = (it & 15) * 2 * 68;
        if constexpr (AHEAD) {
            const float n0 = bp[k], n1 = bp[k + 32];
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            b0 = n0; b1 = n1;
        } else {
            b0 = bp[k]; b1 = bp[k + 32];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int AHEAD>
static void run_lds(int wgs_per_cu, int iters) {
    float *d;
    hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(mfma_lds_loop<AHEAD>, dim3(grid), dim3(256), 0, 0, d, iters / 8, 1.f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_lds_loop<AHEAD>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flops = (double)grid * 4 * iters * 2 * (2.0 * 32 * 32 * 2);
    printf("B fragment from LDS per pair of MFMAs, %s, %d workgroups of 4 waves per CU: %.3f ms  %.1f TFLOP/s  (%.3f of 157.3)\n",
           AHEAD ? "next pair's read issued ahead" : "read -> wait -> MFMAs", wgs_per_cu, best, flops / best / 1e9, flops / best / 1e9 / 157.3);
    hipFree(d);
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

// round 5: K2's inner loop as a TM x TN register tile -- TM A-fragments and TN B-fragments out of LDS per k-step of 2,
// TM * TN MFMAs -- at W workgroups of 4 waves per CU.  2 x 2 is today's tile (one fragment per MFMA); 2 x 4 reads 0.75.
template <int TM, int TN>
__global__ __launch_bounds__(256) void mfma_tile_loop(float *out, int iters, int rnd) {
    __shared__ float as[16 * 260], bs[16 * 132];
    // rnd: operands with random mantissas and signs (what a real GEMM multiplies) instead of ~1.0 -- the data a power-limited
    // clock responds to
    for (int i = threadIdx.x; i < 16 * 260; i += 256) as[i] = rnd ? ((int)((i * 2654435761u + blockIdx.x * 40503u) >> 8) - 8388608) * 1.1920929e-7f : 1.f + i * 1e-6f;
    for (int i = threadIdx.x; i < 16 * 132; i += 256) bs[i] = rnd ? ((int)((i * 2246822519u + blockIdx.x * 9973u) >> 8) - 8388608) * 1.1920929e-7f : 1.f - i * 1e-6f;
    __syncthreads();
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int lane = threadIdx.x & 63, fr = lane & 31, fk = lane >> 5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            float af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = as[(kk + fk) * 260 + a * 32 + fr];
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = bs[(kk + fk) * 132 + b * 32 + fr];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int TM, int TN>
static void run_tile(int wgs_per_cu, int iters, int rnd = 0) {
    float *d;
    hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((mfma_tile_loop<TM, TN>), dim3(grid), dim3(256), 0, 0, d, iters / 8, rnd);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mfma_tile_loop<TM, TN>), dim3(grid), dim3(256), 0, 0, d, iters, rnd);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flops = (double)grid * 4 * iters * 8 * TM * TN * (2.0 * 32 * 32 * 2);
    int nw = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nw, mfma_tile_loop<TM, TN>, 256, 0);
    printf("%sregister tile %d x %d (%d LDS fragments per %d MFMAs), %d workgroups of 4 waves per CU requested (occupancy query: %d): %.3f ms  %.1f TFLOP/s  (%.3f of 157.3)\n",
           rnd ? "[random operands] " : "", TM, TN, TM + TN, TM * TN, wgs_per_cu, nw, best, flops / best / 1e9, flops / best / 1e9 / 157.3);
    hipFree(d);
}

int main(int argc, char **argv) {
    if (argc > 1 && argv[1][0] == 'r') {   // round 5: does the operand data move the sustained rate (power-managed clock)?
        for (int rnd : {0, 1, 0, 1}) { run_tile<2, 4>(4, 4000, rnd); run_tile<1, 4>(4, 8000, rnd); }
        return 0;
    }
    if (argc > 1) {   // round 5: register tiles only
        for (int w : {1, 2, 4}) { run_tile<2, 2>(w, 4000); run_tile<2, 4>(w, 2000); run_tile<1, 4>(w, 4000); run_tile<4, 2>(w, 2000); }
        return 0;
    }
    run<4>(1, 20000);
    run<4>(2, 20000);
    run<4>(4, 20000);    // K2's occupancy (gemm_kernel_occ4) and accumulator count (2 x 2 tiles per wave)
    run<2>(4, 20000);
    run<1>(4, 20000);
    run<4>(4, 200000);   // ~60 ms of MFMA: does the clock sag under sustained load?
    for (int w : {1, 2, 4}) { run_lds<0>(w, 40000); run_lds<1>(w, 40000); }
    run_cls<2, true>(4, 20000);
    run_cls<2, false>(4, 20000);
    run_cls<4, true>(2, 20000);
    run_cls<4, false>(2, 20000);
    return 0;
}
