// lds_tile_probe.hip -- go/no-go probe for an LDS-tiled SpMM on DENSE cells (graphs with locality: a block of a
// community-ordered adjacency holds ~9 edges per destination row and 96 source rows).  Round 2's k1d_probe measured the
// LDS-staged form on the uniform graph (~1 edge per row and tile: 10-20 TB/s, below the L2 gather) and dropped it; this
// probe asks what the same form reaches where a CU really reuses the rows it stages.
//   one 1024-thread workgroup per CU, 32 lane groups x R rows x float4 accumulators (a 512-B slab of R*32 rows, as K1s),
//   source tiles of T rows x 512 B double-buffered in LDS (LDS-DMA copies of tile c+1 while tile c is gathered from),
//   entries (local source row, weight) staged per lane group, rows walked in full batches of 4 + a predicated tail.
//   hipcc --offload-arch=gfx950 -O3 lds_tile_probe.hip -o lds_tile_probe && ./lds_tile_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } \
    } while (0)

constexpr int NT = 1024, GROUP = 32, NGRP = NT / GROUP, NW = NT / 64;
constexpr int T = 96;                 // source rows per tile (48 KB)
constexpr int C = 128;                // staged entries per lane group and cell
constexpr int U = 4;

__device__ __forceinline__ float4 fma4(float w, float4 x, float4 a) {
    a.x = fmaf(x.x, w, a.x); a.y = fmaf(x.y, w, a.y); a.z = fmaf(x.z, w, a.z); a.w = fmaf(x.w, w, a.w);
    return a;
}

struct Args {
    const float4 *x;            // source rows, 32 float4 (512 B) each
    const uint32_t *tile_row0;  // [wgs][cells]: first source row of the cell's tile
    const uint32_t *off;        // [wgs][cells][NGRP*R + 1]: entry offsets of the workgroup's rows inside the cell
    const uint2 *ent;           // (local source row, weight bits), cell-major per workgroup
    const uint64_t *cell_base;  // [wgs][cells]: first entry of the cell
    float4 *out;                // [wgs][NGRP*R][32]
    int cells;
    int fill;                   // 0: tiles loaded once (inner loop alone), 1: every cell loads its tile
};

template <int R>
__global__ __launch_bounds__(NT) void lds_tile_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *tile = reinterpret_cast<float4 *>(smem);                         // [2][T][32]
    uint2 *stage = reinterpret_cast<uint2 *>(smem + 2 * T * 512);            // [NGRP][C]
    uint32_t *offl = reinterpret_cast<uint32_t *>(smem + 2 * T * 512 + NGRP * C * 8);   // [NGRP*R + 1]
    constexpr int RW = NGRP * R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, g = threadIdx.x >> 5;
    const uint32_t wg = blockIdx.x;
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    // tile copy: wave w copies rows [w * T/NW, (w+1) * T/NW), two rows (64 lanes x 16 B) per LDS-DMA instruction
    auto copy_tile = [&](int c, int buf) {
        const uint32_t row0 = a.tile_row0[(size_t)wg * a.cells + c];
#pragma unroll
        for (int j = 0; j < T / NW / 2; ++j) {
            const int r = wave * (T / NW) + 2 * j;
            __builtin_amdgcn_global_load_lds((gptr_t)(a.x + (size_t)(row0 + r) * 32 + lane), (lptr_t)(tile + ((size_t)buf * T + r) * 32), 16, 0, 0);
        }
    };
    copy_tile(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    for (int c = 0; c < a.cells; ++c) {
        const int buf = a.fill ? (c & 1) : 0;
        if (a.fill && c + 1 < a.cells) copy_tile(c + 1, buf ^ 1);           // in flight during this cell
        // offsets + entries of this cell
        const uint32_t *off = a.off + ((size_t)wg * a.cells + c) * (RW + 1);
        for (int i = threadIdx.x; i <= RW; i += NT) offl[i] = off[i];
        const uint64_t base = a.cell_base[(size_t)wg * a.cells + c];
        const uint32_t o0 = off[g * R], oR = off[g * R + R];
        for (uint32_t e = o0 + li; e < oR && e - o0 < C; e += GROUP) stage[g * C + (e - o0)] = a.ent[base + e];
        __syncthreads();
        const float4 *tl = tile + (size_t)buf * T * 32 + li;
        const uint2 *st = stage + g * C;
        const uint32_t *ol = offl + g * R;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t lo = ol[r] - o0, hi = min(ol[r + 1] - o0, (uint32_t)C);
            uint32_t e = lo;
            for (; e + U <= hi; e += U) {
                uint2 en[U];
                float4 xv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) en[u] = st[e + u];
#pragma unroll
                for (int u = 0; u < U; ++u) xv[u] = tl[en[u].x * 32];
#pragma unroll
                for (int u = 0; u < U; ++u) acc[r] = fma4(__uint_as_float(en[u].y), xv[u], acc[r]);
            }
            for (; e < hi; ++e) {
                const uint2 en = st[e];
                acc[r] = fma4(__uint_as_float(en.y), tl[en.x * 32], acc[r]);
            }
        }
        if (a.fill) __builtin_amdgcn_s_waitcnt(0x0f70);                      // the next tile has landed
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < R; ++r) a.out[((size_t)wg * RW + g * R + r) * 32 + li] = acc[r];
}

template <int R>
static void run(double lambda, int cells, int fill, const float4 *dx, uint32_t xrows) {
    constexpr int RW = NGRP * R;
    const int wgs = 256;
    std::mt19937_64 rng(7);
    std::poisson_distribution<int> pois(lambda);
    std::vector<uint32_t> off((size_t)wgs * cells * (RW + 1)), row0((size_t)wgs * cells);
    std::vector<uint64_t> base((size_t)wgs * cells);
    std::vector<uint2> ent;
    ent.reserve((size_t)(wgs * cells * RW * lambda * 1.1));
    uint64_t edges = 0, dropped = 0;
    for (int w = 0; w < wgs; ++w)
        for (int c = 0; c < cells; ++c) {
            const size_t k = (size_t)w * cells + c;
            row0[k] = (uint32_t)(rng() % (xrows - T));
            base[k] = ent.size();
            uint32_t *o = off.data() + k * (RW + 1);
            uint32_t run_ = 0;
            for (int g = 0; g < NGRP; ++g) {
                uint32_t in_group = 0;
                for (int r = 0; r < R; ++r) {
                    o[g * R + r] = run_;
                    int n = pois(rng);
                    if (in_group + n > C) { dropped += in_group + n - C; n = C - in_group; }
                    for (int i = 0; i < n; ++i) {
                        const float wv = 0.001f * (float)(rng() % 1000);
                        ent.push_back(make_uint2((uint32_t)(rng() % T), *reinterpret_cast<const uint32_t *>(&wv)));
                    }
                    in_group += n; run_ += n; edges += n;
                }
            }
            o[RW] = run_;
        }
    uint32_t *d_off, *d_row0; uint64_t *d_base; uint2 *d_ent; float4 *d_out;
    CK(hipMalloc(&d_off, off.size() * 4)); CK(hipMalloc(&d_row0, row0.size() * 4)); CK(hipMalloc(&d_base, base.size() * 8));
    CK(hipMalloc(&d_ent, ent.size() * 8)); CK(hipMalloc(&d_out, (size_t)wgs * RW * 512));
    CK(hipMemcpy(d_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_row0, row0.data(), row0.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_base, base.data(), base.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ent, ent.data(), ent.size() * 8, hipMemcpyHostToDevice));
    Args a{dx, d_row0, d_off, d_ent, d_base, d_out, cells, fill};
    const size_t lds = 2 * T * 512 + NGRP * C * 8 + (RW + 1) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(lds_tile_kernel<R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(lds_tile_kernel<R>, dim3(wgs), dim3(NT), lds, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(lds_tile_kernel<R>, dim3(wgs), dim3(NT), lds, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double tbs = (double)edges * 512 / (ms * 1e-3) / 1e12;
    printf("R=%d lambda=%.1f edges per (row, tile), %d cells, fill=%d: %.3f ms  %.1f Gedge/s  LDS gather %.2f TB/s (%.1f B/clk/CU @2.4GHz)  tile fill traffic %.2f TB/s  per cell %.2f us  (entries dropped by the stage cap: %.2f %%)\n",
           R, lambda, cells, fill, ms, edges / (ms * 1e-3) / 1e9, tbs, tbs * 1e12 / 256 / 2.4e9,
           fill ? (double)wgs * cells * T * 512 / (ms * 1e-3) / 1e12 : 0.0, ms * 1e3 / cells, 100.0 * dropped / (double)(edges + dropped));
    CK(hipFree(d_off)); CK(hipFree(d_row0)); CK(hipFree(d_base)); CK(hipFree(d_ent)); CK(hipFree(d_out));
}

int main() {
    const uint32_t xrows = 232965;
    float4 *dx;
    CK(hipMalloc(&dx, (size_t)xrows * 512));
    CK(hipMemset(dx, 0, (size_t)xrows * 512));
    // community-ordered Reddit-size graph (bench.py --graph community): 418 internal edges per row over 4660 rows -> 8.6 per 96-row tile
    for (int fill = 0; fill < 2; ++fill) {
        run<10>(8.6, 48, fill, dx, xrows);
        run<8>(8.6, 48, fill, dx, xrows);
    }
    run<10>(4.3, 48, 1, dx, xrows);
    run<10>(2.0, 48, 1, dx, xrows);
    run<10>(1.0, 48, 1, dx, xrows);     // ~ the uniform graph's density per tile of this size x 5: where the form stops paying
    return 0;
}
