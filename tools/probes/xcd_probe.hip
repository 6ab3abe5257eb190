// xcd_probe.hip -- which XCD does workgroup i land on, and are workgroup-scope atomics
// coherent across the CUs of one XCD?   hipcc --offload-arch=gfx950 -O2 xcd_probe.hip -o xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned *xcc, unsigned *cnt_wg, unsigned *cnt_agent, unsigned *seen_wg, unsigned *seen_agent,
                      unsigned spin) {
    // HW_REG_XCC_ID = 20, bits [3:0]
    const unsigned id = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
    if (threadIdx.x == 0) {
        xcc[blockIdx.x] = id;
        __hip_atomic_fetch_add(cnt_wg + id * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(cnt_agent + id * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (unsigned i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
        seen_wg[blockIdx.x] = __hip_atomic_fetch_add(cnt_wg + id * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        seen_agent[blockIdx.x] = __hip_atomic_load(cnt_agent + id * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char **argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 2048;
    unsigned *xcc, *cw, *ca, *sw, *sa;
    hipMalloc(&xcc, grid * 4); hipMalloc(&sw, grid * 4); hipMalloc(&sa, grid * 4);
    hipMalloc(&cw, 16 * 32 * 4); hipMalloc(&ca, 16 * 32 * 4);
    hipMemset(cw, 0, 16 * 32 * 4); hipMemset(ca, 0, 16 * 32 * 4);
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, xcc, cw, ca, sw, sa, 200u);
    hipDeviceSynchronize();
    std::vector<unsigned> h(grid), hw(grid), ha(grid);
    hipMemcpy(h.data(), xcc, grid * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hw.data(), sw, grid * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ha.data(), sa, grid * 4, hipMemcpyDeviceToHost);
    int match = 0, hist[16] = {0};
    for (int i = 0; i < grid; ++i) { match += (h[i] == (unsigned)(i & 7)); hist[h[i] & 15]++; }
    printf("grid %d: blocks with xcc_id == blockIdx&7: %d\nper-XCD blocks:", grid, match);
    for (int x = 0; x < 8; ++x) printf(" %d", hist[x]);
    printf("\nfirst 24 xcc ids:");
    for (int i = 0; i < 24 && i < grid; ++i) printf(" %u", h[i]);
    unsigned mw = 0, ma = 0;
    for (int i = 0; i < grid; ++i) { mw = hw[i] > mw ? hw[i] : mw; ma = ha[i] > ma ? ha[i] : ma; }
    printf("\nmax count seen: workgroup-scope %u, agent-scope %u (blocks per XCD %d)\n", mw, ma, hist[0]);
    return 0;
}
