// spmm_lab.hip -- bench for compile-time variants of the PRODUCT SpMM kernels at Reddit scale, outside the C-ABI.
// It includes the product translation unit itself (dorylus_amd/csrc/spmm.hip: K1 and K1s), so an experiment is an edit
// of -- or a -D switch for -- the one kernel source; there is no second copy to drift.  (Rounds 2-3 kept five generations
// of K1s in this file; they are in the history: git log -- tools/probes/spmm_lab.hip.)  GPU box only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude [-DSWEEP_DMA_AUX=0 ...] -c tools/probes/spmm_lab.hip -o /tmp/spmm_lab.o && \
//   hipcc --offload-arch=gfx950 /tmp/spmm_lab.o dorylus_amd/host/sweep_deal.o -o tools/probes/spmm_lab
//   tools/probes/spmm_lab [F=602] [deg=492] [window_rows=N] [rows_per_group=10] [window_kb=2432] [flags=0] [loader=1]
//     window_rows < N draws every source from [0, window_rows): the L2-resident ceiling of the same kernel
//     flags: 8 = no gates (SweepArgs::flags)
#include "../../dorylus_amd/csrc/spmm.hip"

#include <cstdio>
#include <cstdlib>

using namespace dory;

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e__ = (x);                                                                  \
        if (e__ != hipSuccess) {                                                               \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e__), __FILE__, __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

static double rel_err(const std::vector<float> &a, const std::vector<float> &b) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        num = std::max(num, (double)std::fabs(a[i] - b[i]));
        den = std::max(den, (double)std::fabs(b[i]));
    }
    return num / std::max(den, 1e-30);
}

template <class F>
static float time_ms(F f, int iters = 3) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f();   // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char **argv) {
    const uint32_t N = 232965;
    const uint32_t F = argc > 1 ? atoi(argv[1]) : 602;
    const uint32_t deg = argc > 2 ? atoi(argv[2]) : 492;
    const uint32_t win = argc > 3 && atoi(argv[3]) > 0 ? atoi(argv[3]) : N;
    const int R = argc > 4 ? atoi(argv[4]) : 10;
    const uint64_t window_kb = argc > 5 ? atoi(argv[5]) : 2432;
    const uint32_t flags = argc > 6 ? atoi(argv[6]) : 0;
    const bool loader = argc > 7 ? atoi(argv[7]) != 0 : true;
    const uint32_t ld = (F + 31) & ~31u;
    const uint64_t E = (uint64_t)N * deg;
    printf("N=%u deg=%u E=%llu F=%u ld=%u sources from [0,%u) R=%d window %llu KB flags %u loader %d\n", N, deg,
           (unsigned long long)E, F, ld, win, R, (unsigned long long)window_kb, flags, (int)loader);
    std::vector<uint64_t> ptr(N + 1);
    std::vector<uint32_t> idx(E);
    std::vector<float> val(E);
    uint64_t st = 88172645463325252ull;
    for (uint32_t v = 0; v <= N; ++v) ptr[v] = (uint64_t)v * deg;
    for (uint64_t e = 0; e < E; ++e) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        idx[e] = (uint32_t)((st >> 11) % win);
        val[e] = (float)((st >> 40) & 0xFFFF) * (0.01f / 65536.f);
    }
    uint64_t *d_ptr; uint32_t *d_idx; float *d_val, *d_x, *d_out, *d_ref, *d_self;
    CK(hipMalloc(&d_ptr, (N + 1) * 8));
    CK(hipMalloc(&d_idx, E * 4));
    CK(hipMalloc(&d_val, E * 4));
    CK(hipMalloc(&d_x, (size_t)N * ld * 4));
    CK(hipMalloc(&d_out, (size_t)N * ld * 4));
    CK(hipMalloc(&d_ref, (size_t)N * ld * 4));
    CK(hipMalloc(&d_self, N * 4));
    CK(hipMemcpy(d_ptr, ptr.data(), (N + 1) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_idx, idx.data(), E * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_val, val.data(), E * 4, hipMemcpyHostToDevice));
    {
        std::vector<float> hx((size_t)N * ld, 0.f), hs(N);
        for (uint32_t v = 0; v < N; ++v) {
            hs[v] = 1.f / (deg + 1);
            for (uint32_t c = 0; c < F; ++c) {
                st ^= st << 13; st ^= st >> 7; st ^= st << 17;
                hx[(size_t)v * ld + c] = (float)((st >> 40) & 0xFFFF) * (2.f / 65536.f) - 1.f;
            }
        }
        CK(hipMemcpy(d_x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_self, hs.data(), N * 4, hipMemcpyHostToDevice));
    }
    SpmmArgs a{};
    a.N = N; a.F = F; a.ld = ld; a.ptr = d_ptr; a.idx = d_idx; a.val = d_val; a.self_scale = d_self; a.self_mode = 1;
    a.xl = d_x; a.xg = nullptr; a.out = d_ref;
    const double gather = (double)E * ld * 4;
    const float t_ref = time_ms([&] { CK(launch_spmm(a, 0, 64, 0)); }, 1);     // reference: K1 row kernel
    printf("K1 (slab 64): %.3f ms  gather %.2f TB/s\n", t_ref, gather / t_ref / 1e9);
    std::vector<float> href((size_t)N * ld), hout((size_t)N * ld);
    CK(hipMemcpy(href.data(), d_ref, href.size() * 4, hipMemcpyDeviceToHost));
    a.out = d_out;

    // K1s: the product's layout and launcher
    BlockedAdj S{};
    CK(build_blocked_sweep(d_ptr, d_idx, d_val, N, N, E, 0, 512, window_kb << 10, R, &S, 0, 3, 32, loader ? 3u : 0u));
    uint32_t *d_done, *d_stat;
    const size_t scratch = sweep_scratch_bytes(S, ld, 32, 32, S.nb, R);
    CK(hipMalloc(&d_done, scratch));
    CK(hipMalloc(&d_stat, SWEEP_STAT_WORDS * sizeof(uint32_t)));
    CK(hipMemset(d_stat, 0, SWEEP_STAT_WORDS * sizeof(uint32_t)));
    float *d_split = nullptr;
    if (S.nslots) CK(hipMalloc(&d_split, (size_t)S.nslots * ld * 4));
    SweepCtl ctl;
    ctl.force_r = R;
    ctl.stat = d_stat;
    ctl.loader = loader;
    CK(hipMemset(d_out, 0, (size_t)N * ld * 4));
    const float t = time_ms([&] {
        CK(launch_spmm_sweep(a, S, 32, nullptr, 32, 0, S.nb, d_done, 0, ctl, flags, d_split));
        CK(launch_spmm_sweep_combine(a, S, nullptr, d_split, 0));
    });
    CK(hipMemcpy(hout.data(), d_out, hout.size() * 4, hipMemcpyDeviceToHost));
    uint32_t hstat[SWEEP_STAT_WORDS];
    CK(hipMemcpy(hstat, d_stat, sizeof(hstat), hipMemcpyDeviceToHost));
    printf("K1s nb=%u window %.2f MB R=%d loader %d flags %u: %.3f ms  gather %.2f TB/s  err vs K1 %.2e  gate timeouts %u  ungated %u  launches %u\n",
           S.nb, (double)window_kb / 1024.0, R, (int)loader, flags, t, gather / t / 1e9, rel_err(hout, href), hstat[0], hstat[2], hstat[4]);
    free_blocked(&S);
    return 0;
}
