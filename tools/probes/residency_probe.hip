// residency_probe.hip -- how many workgroups of T threads, V registers per lane and L bytes of LDS does a CU of gfx950
// really hold at once, and which SIMD do a workgroup's waves land on?  (Round 4: K1s with two 640-thread workgroups per CU
// timed out at its gates; a 10-wave workgroup puts 3+3+2+2 waves on the four SIMDs.)
//   hipcc --offload-arch=gfx950 -O2 residency_probe.hip -o residency_probe;  ./residency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

template <int T, int WPE, int VREG>
__global__ __launch_bounds__(T, WPE) void occupy(unsigned long long ticks, unsigned *hwid, unsigned long long *t_start, unsigned *lds_sink) {
    extern __shared__ unsigned lds[];
    if constexpr (VREG == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if constexpr (VREG == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
    if constexpr (VREG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    const unsigned long long t0 = wall_clock64();
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        // HW_REG_HW_ID (4): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] ...; XCC_ID (20)
        const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
        const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
        hwid[(size_t)blockIdx.x * (T / 64) + wave] = (hw & 0xFFFFFFu) | (xcc << 24);
        if (wave == 0) t_start[blockIdx.x] = t0;
    }
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (ticks == ~0ull) { lds[threadIdx.x] = 1; lds_sink[0] = lds[0]; }
}

template <int T, int WPE, int VREG>
static void run(int per_cu_expected, size_t lds_bytes, const char *what) {
    const int grid = 8 * 32 * per_cu_expected;      // exactly what should be resident at once
    const int W = T / 64;
    unsigned *hwid; unsigned long long *ts;
    hipMalloc(&hwid, (size_t)grid * W * 4); hipMalloc(&ts, (size_t)grid * 8);
    hipFuncSetAttribute(reinterpret_cast<const void *>(occupy<T, WPE, VREG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL((occupy<T, WPE, VREG>), dim3(grid), dim3(T), lds_bytes, 0, 20000ull /* 200 us */, hwid, ts, (unsigned *)nullptr);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", what); return; }
    std::vector<unsigned> h((size_t)grid * W); std::vector<unsigned long long> t(grid);
    hipMemcpy(h.data(), hwid, h.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(t.data(), ts, t.size() * 8, hipMemcpyDeviceToHost);
    const unsigned long long tmin = *std::min_element(t.begin(), t.end());
    int early = 0;
    for (int i = 0; i < grid; ++i) early += (t[i] - tmin) < 10000ull;   // started within 100 us of the first: resident together
    // waves per SIMD of each workgroup, and of the first CU seen
    int simd_hist[4] = {0, 0, 0, 0}, max_per_simd_wg = 0;
    for (int i = 0; i < grid; ++i) {
        int c[4] = {0, 0, 0, 0};
        for (int w = 0; w < W; ++w) c[(h[(size_t)i * W + w] >> 4) & 3]++;
        for (int s = 0; s < 4; ++s) max_per_simd_wg = std::max(max_per_simd_wg, c[s]);
        if (i == 0) for (int s = 0; s < 4; ++s) simd_hist[s] = c[s];
    }
    printf("%-58s grid %4d: resident together %4d (%.2f per CU); workgroup 0 waves per SIMD %d %d %d %d; most waves of one workgroup on a SIMD: %d\n",
           what, grid, early, early / 256.0, simd_hist[0], simd_hist[1], simd_hist[2], simd_hist[3], max_per_simd_wg);
    hipFree(hwid); hipFree(ts);
}

int main() {
    run<1024, 4, 128>(1, 100 << 10, "1024 threads, 128 VGPRs, 100 KB LDS (K1s today), 1 per CU");
    run<640, 5, 96>(2, 64 << 10, "640 threads, 96 VGPRs, 64 KB LDS, 2 per CU");
    run<640, 5, 96>(2, 16 << 10, "640 threads, 96 VGPRs, 16 KB LDS, 2 per CU");
    run<640, 6, 80>(2, 64 << 10, "640 threads, 80 VGPRs, 64 KB LDS, 2 per CU");
    run<512, 5, 96>(2, 64 << 10, "512 threads, 96 VGPRs, 64 KB LDS, 2 per CU");
    run<768, 3, 128>(1, 144 << 10, "768 threads, 128 (<=168) VGPRs, 144 KB LDS, 1 per CU");
    run<256, 5, 96>(5, 24 << 10, "256 threads, 96 VGPRs, 24 KB LDS, 5 per CU");
    run<320, 5, 96>(4, 32 << 10, "320 threads, 96 VGPRs, 32 KB LDS, 4 per CU");
    return 0;
}
