// k1d_probe.hip -- go/no-go probe for an LDS-tiled SpMM ("K1d"): a workgroup keeps R destination rows per 16-lane
// group in registers (64-float slab), stages a block of BS source rows (256 B each) in LDS and walks a tagged entry
// stream (src_in_block | slot<<9) per group.  Measures edges/s per CU for the inner loop alone and with the tile fill.
// Synthetic Poisson(lambda) segment lengths = what a uniform random graph gives.  GPU box only:
//   hipcc --offload-arch=gfx950 -O3 -o k1d_probe k1d_probe.hip && ./k1d_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);       \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

constexpr int BS = 512;      // source rows per LDS tile
constexpr int W4 = 16;       // float4 per row slab
constexpr int NT = 1024;
constexpr int NGRP = NT / 16;
constexpr int ENT_CAP = 4096;
constexpr int XT_BYTES = BS * W4 * 16;                  // 131072
constexpr int ENT_OFF = XT_BYTES;                       // u16 tags
constexpr int VAL_OFF = ENT_OFF + ENT_CAP * 2;          // f32 vals
constexpr int HDR_OFF = VAL_OFF + ENT_CAP * 4;
constexpr int SMEM = HDR_OFF + NGRP * 2;

__device__ __forceinline__ float4 fma4(float w, float4 x, float4 a) {
    a.x = fmaf(x.x, w, a.x);
    a.y = fmaf(x.y, w, a.y);
    a.z = fmaf(x.z, w, a.z);
    a.w = fmaf(x.w, w, a.w);
    return a;
}
__device__ __forceinline__ float4 add4(float4 x, float4 a) {
    a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
    return a;
}

template <int R, bool UNIT, int PIPE>
__global__ __launch_bounds__(NT) void k1d_probe(const float4 *__restrict__ x, uint32_t xrows,
                                                const uint32_t *__restrict__ step_off, const uint16_t *__restrict__ tags,
                                                const float *__restrict__ vals, const uint16_t *__restrict__ hdrs,
                                                int nsteps, int nreg, int do_fill, float4 *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *xt = reinterpret_cast<float4 *>(smem);
    uint16_t *ent = reinterpret_cast<uint16_t *>(smem + ENT_OFF);
    float *ev = reinterpret_cast<float *>(smem + VAL_OFF);
    uint16_t *hdr = reinterpret_cast<uint16_t *>(smem + HDR_OFF);
    const int tid = threadIdx.x, l16 = tid & 15, g = tid >> 4;
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!do_fill) {
        for (int k = 0; k < 8; ++k) xt[k * NT + tid] = x[k * NT + tid];
    }
    for (int s = 0; s < nsteps; ++s) {
        __syncthreads();
        if (do_fill) {
            const size_t base = ((size_t)(blockIdx.x >> 3) * 3 + s) % (xrows / BS) * BS * W4;
#pragma unroll
            for (int k = 0; k < 8; ++k) xt[k * NT + tid] = x[base + k * NT + tid];
        }
        const int reg = (s + blockIdx.x) % nreg;
        const uint32_t o0 = step_off[reg], n = step_off[reg + 1] - o0;
        for (uint32_t i = tid; i < (n + 1) / 2; i += NT)
            reinterpret_cast<uint32_t *>(ent)[i] = reinterpret_cast<const uint32_t *>(tags + o0)[i];
        if (!UNIT)
            for (uint32_t i = tid; i < n; i += NT) ev[i] = vals[o0 + i];
        if (tid < NGRP) hdr[tid] = hdrs[reg * NGRP + tid];
        __syncthreads();
        uint32_t pos = hdr[g];
        if constexpr (PIPE == 0) {
            uint32_t e = ent[pos];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                while ((e >> 9) == (uint32_t)r) {
                    const float4 xv = xt[(e & 511u) * W4 + l16];
                    if constexpr (UNIT) acc[r] = add4(xv, acc[r]);
                    else acc[r] = fma4(ev[pos], xv, acc[r]);
                    ++pos;
                    e = ent[pos];
                }
            }
        } else if constexpr (PIPE == 1) {
            uint32_t e = ent[pos];
            float4 xv = xt[(e & 511u) * W4 + l16];
            float w = UNIT ? 1.f : ev[pos];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                while ((e >> 9) == (uint32_t)r) {
                    const uint32_t e2 = ent[pos + 1];
                    const float4 x2 = xt[(e2 & 511u) * W4 + l16];
                    const float w2 = UNIT ? 1.f : ev[pos + 1];
                    if constexpr (UNIT) acc[r] = add4(xv, acc[r]);
                    else acc[r] = fma4(w, xv, acc[r]);
                    ++pos;
                    e = e2; xv = x2; w = w2;
                }
            }
        } else {
            uint32_t e = ent[pos], e1 = ent[pos + 1];
            float4 xv = xt[(e & 511u) * W4 + l16], x1 = xt[(e1 & 511u) * W4 + l16];
            float w = UNIT ? 1.f : ev[pos], w1 = UNIT ? 1.f : ev[pos + 1];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                while ((e >> 9) == (uint32_t)r) {
                    const uint32_t e2 = ent[pos + 2];
                    const float4 x2 = xt[(e2 & 511u) * W4 + l16];
                    const float w2 = UNIT ? 1.f : ev[pos + 2];
                    if constexpr (UNIT) acc[r] = add4(xv, acc[r]);
                    else acc[r] = fma4(w, xv, acc[r]);
                    ++pos;
                    e = e1; xv = x1; w = w1;
                    e1 = e2; x1 = x2; w1 = w2;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) out[((size_t)blockIdx.x * NGRP * R + g * R + r) * W4 + l16] = acc[r];
}

struct Streams {
    std::vector<uint32_t> step_off;
    std::vector<uint16_t> tags, hdrs;
    std::vector<float> vals;
    double avg_edges;
};

static Streams make_streams(int R, double lambda, int nreg, unsigned seed) {
    Streams S;
    std::mt19937 rng(seed);
    std::poisson_distribution<int> pois(lambda);
    std::uniform_int_distribution<int> src(0, BS - 1);
    std::uniform_real_distribution<float> uf(0.f, 0.01f);
    S.step_off.push_back(0);
    uint64_t edges = 0;
    for (int reg = 0; reg < nreg; ++reg) {
        const size_t base = S.tags.size();
        for (int g = 0; g < NGRP; ++g) {
            S.hdrs.push_back((uint16_t)(S.tags.size() - base));
            for (int r = 0; r < R; ++r) {
                const int len = pois(rng);
                for (int j = 0; j < len; ++j) {
                    S.tags.push_back((uint16_t)(src(rng) | (r << 9)));
                    S.vals.push_back(uf(rng));
                    ++edges;
                }
            }
            for (int t = 0; t < 3; ++t) { S.tags.push_back((uint16_t)(31u << 9)); S.vals.push_back(0.f); }
        }
        if (S.tags.size() & 1) { S.tags.push_back((uint16_t)(31u << 9)); S.vals.push_back(0.f); }
        if (S.tags.size() - base > ENT_CAP) { printf("region too large\n"); exit(1); }
        S.step_off.push_back((uint32_t)S.tags.size());
    }
    S.avg_edges = (double)edges / nreg;
    return S;
}

template <int R, bool UNIT, int PIPE>
static void run(double lambda, int do_fill, const float4 *dx, uint32_t xrows, float4 *dout) {
    const int nreg = 64, nsteps = 200, nwg = 512;
    Streams S = make_streams(R, lambda, nreg, 7);
    uint32_t *d_off; uint16_t *d_tags, *d_hdr; float *d_vals;
    CK(hipMalloc(&d_off, S.step_off.size() * 4));
    CK(hipMalloc(&d_tags, S.tags.size() * 2 + 64));
    CK(hipMalloc(&d_vals, S.vals.size() * 4 + 64));
    CK(hipMalloc(&d_hdr, S.hdrs.size() * 2));
    CK(hipMemcpy(d_off, S.step_off.data(), S.step_off.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tags, S.tags.data(), S.tags.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_vals, S.vals.data(), S.vals.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_hdr, S.hdrs.data(), S.hdrs.size() * 2, hipMemcpyHostToDevice));
    auto kern = k1d_probe<R, UNIT, PIPE>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(NT), SMEM, 0, dx, xrows, d_off, d_tags, d_vals, d_hdr, nsteps, nreg,
                           do_fill, dout);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double edges = (double)nwg * nsteps * S.avg_edges;
    const double t = ms * 1e-3;
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern)));
    printf("R=%2d unit=%d pipe=%d lambda=%.2f fill=%d: %8.3f ms  %7.2f Gedge/s  LDS-read %6.2f TB/s  (%.1f B/clk/CU @2.1GHz)  "
           "edges/step %.0f  vgpr %d spill %d\n",
           R, (int)UNIT, PIPE, lambda, do_fill, ms, edges / t / 1e9, edges * 256 / t / 1e12,
           edges * 256 / t / 256 / 2.1e9, S.avg_edges, fa.numRegs, (int)fa.localSizeBytes);
    fflush(stdout);
    CK(hipFree(d_off)); CK(hipFree(d_tags)); CK(hipFree(d_vals)); CK(hipFree(d_hdr));
}

int main() {
    const uint32_t xrows = 1u << 18;   // 64 MB slab column: fills come from L2/MALL mix
    float4 *dx, *dout;
    CK(hipMalloc(&dx, (size_t)xrows * W4 * 16));
    CK(hipMemset(dx, 0, (size_t)xrows * W4 * 16));
    CK(hipMalloc(&dout, (size_t)512 * NGRP * 24 * W4 * 16));
    const double lam = BS * 492.0 / 232965.0;
    for (int fill = 0; fill < 2; ++fill) {
        run<24, true, 0>(lam, fill, dx, xrows, dout);
        run<24, true, 1>(lam, fill, dx, xrows, dout);
        run<24, true, 2>(lam, fill, dx, xrows, dout);
        run<24, false, 0>(lam, fill, dx, xrows, dout);
        run<24, false, 1>(lam, fill, dx, xrows, dout);
        run<24, false, 2>(lam, fill, dx, xrows, dout);
        run<16, true, 1>(lam, fill, dx, xrows, dout);
        run<16, false, 1>(lam, fill, dx, xrows, dout);
        run<20, false, 1>(lam, fill, dx, xrows, dout);
    }
    run<24, false, 1>(2 * lam, 0, dx, xrows, dout);
    run<12, false, 1>(4 * lam, 0, dx, xrows, dout);
    run<24, false, 1>(0.5 * lam, 0, dx, xrows, dout);
    return 0;
}
