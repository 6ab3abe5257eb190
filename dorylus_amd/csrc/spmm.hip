// spmm.hip -- K1: CSC/CSR SpMM with fused self term, fp32, for gfx950.
//
// Replaces Engine::aggregateGCN / aggregateGAT (reference
// src/graph-server/engine/ops/gcn_ops.cpp:130-191, gat_ops.cpp:173-243) and the
// cuSPARSE SpMM + transpose + dgmm + add chain of
// GPU-Computation/comp_unit.cu:48-91,150,159.
//
//   out[v,:] = self[v] * xl[v,:]  +  sum_{e in [ptr[v],ptr[v+1])} val[e] * row(idx[e])[:]
//   row(i) = xl[i] if i < N else xg[i-N]          (ghost ids are N+k)
//
// Mapping (HBM/fabric-bound gather, no MFMA):
//   * one GROUP of lanes (8/16/32/64) owns one output row; each lane keeps
//     CHUNKS float4 accumulators, so a 64-lane group covers up to 256*CHUNKS
//     floats with 16-B coalesced loads (a 608-float row = 19 full 128-B lines).
//   * the group loads GROUP (idx,val) pairs at a time with one coalesced load
//     each and broadcasts them (readlane for 64-lane groups -> scalar row base,
//     ds_bpermute otherwise); 4 source rows are in flight per group.
//   * products are accumulated in edge order with fmaf -- the same order of
//     additions as the reference loop, so results differ from the CPU path only
//     by fma contraction.
//   * optional feature slabs (gridDim.y): each slab re-reads the index arrays but
//     keeps the gathered working set N x slab x 4 B small enough for the 256 MB
//     Infinity Cache.
//   * optional row schedule (longest row first) for skewed degree distributions.
#include "ctx.hpp"

namespace dory {

template <int GROUP>
__device__ __forceinline__ uint32_t bcast_u32(uint32_t v, int j) {
    if constexpr (GROUP == 64) {
        return (uint32_t)__builtin_amdgcn_readlane((int)v, j);
    } else {
        return (uint32_t)__shfl((int)v, j, GROUP);
    }
}
template <int GROUP>
__device__ __forceinline__ float bcast_f32(float v, int j) {
    if constexpr (GROUP == 64) {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
    } else {
        return __shfl(v, j, GROUP);
    }
}

__device__ __forceinline__ float4 fma4(float w, float4 x, float4 a) {
    a.x = fmaf(x.x, w, a.x);
    a.y = fmaf(x.y, w, a.y);
    a.z = fmaf(x.z, w, a.z);
    a.w = fmaf(x.w, w, a.w);
    return a;
}

template <int GROUP, int CHUNKS>
__global__ __launch_bounds__(256) void spmm_rows_kernel(SpmmArgs a) {
    constexpr int RPW = 64 / GROUP;      // rows per wave
    constexpr int RPB = 4 * RPW;         // rows per 256-thread block
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane % GROUP;
    const int gi = lane / GROUP;
    const uint32_t rid = blockIdx.x * RPB + wave * RPW + gi;
    const bool row_ok = rid < a.N;
    uint32_t v = row_ok ? (a.order ? a.order[rid] : rid) : 0;
    if constexpr (GROUP == 64)  // one row per wave: make it provably wave-uniform (scalar loads, SGPR row base)
        v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);

    const uint32_t nchunk = a.ld >> 2;                       // float4 per row
    const uint32_t c0 = blockIdx.y * (GROUP * CHUNKS) + li;  // first chunk of this lane
    bool act[CHUNKS];
    uint32_t col[CHUNKS];  // inactive lanes gather chunk 0 (valid memory) and never store
    float4 acc[CHUNKS];
#pragma unroll
    for (int k = 0; k < CHUNKS; ++k) {
        act[k] = row_ok && (c0 + GROUP * k) < nchunk;
        col[k] = act[k] ? c0 + GROUP * k : 0;
        acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    const float4 *xl4 = reinterpret_cast<const float4 *>(a.xl);
    const float4 *xg4 = reinterpret_cast<const float4 *>(a.xg);

    if (a.self_mode != 0) {
        const float sc = (a.self_mode == 1 && row_ok) ? a.self_scale[v] : 1.f;
#pragma unroll
        for (int k = 0; k < CHUNKS; ++k)
            if (act[k]) {
                float4 x = xl4[(size_t)v * nchunk + col[k]];
                acc[k] = make_float4(x.x * sc, x.y * sc, x.z * sc, x.w * sc);
            }
    }

    uint64_t e = row_ok ? a.ptr[v] : 0;
    const uint64_t end = row_ok ? a.ptr[v + 1] : 0;
    // GROUP == 64: the loop is wave-uniform (one row per wave).
    while (e < end) {
        const int n = (end - e) < (uint64_t)GROUP ? (int)(end - e) : GROUP;
        uint32_t my_idx = 0;
        float my_val = 0.f;
        if (li < n) {
            my_idx = a.idx[e + li];
            my_val = a.val[e + li];
        }
        int j = 0;
        for (; j + 4 <= n; j += 4) {
            float4 x[4][CHUNKS];
            float w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t s = bcast_u32<GROUP>(my_idx, j + u);
                w[u] = bcast_f32<GROUP>(my_val, j + u);
                const float4 *row = s < a.N ? xl4 + (size_t)s * nchunk
                                            : xg4 + (size_t)(s - a.N) * nchunk;
#pragma unroll
                for (int k = 0; k < CHUNKS; ++k)
                    x[u][k] = row[col[k]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < CHUNKS; ++k) acc[k] = fma4(w[u], x[u][k], acc[k]);
        }
        for (; j < n; ++j) {
            const uint32_t s = bcast_u32<GROUP>(my_idx, j);
            const float w = bcast_f32<GROUP>(my_val, j);
            const float4 *row = s < a.N ? xl4 + (size_t)s * nchunk
                                        : xg4 + (size_t)(s - a.N) * nchunk;
#pragma unroll
            for (int k = 0; k < CHUNKS; ++k)
                acc[k] = fma4(w, row[col[k]], acc[k]);
        }
        e += n;
    }

    float4 *out4 = reinterpret_cast<float4 *>(a.out);
#pragma unroll
    for (int k = 0; k < CHUNKS; ++k)
        if (act[k]) {
            const size_t o = (size_t)v * nchunk + col[k];
            if (a.accumulate) {
                float4 p = out4[o];
                acc[k].x += p.x; acc[k].y += p.y; acc[k].z += p.z; acc[k].w += p.w;
            }
            out4[o] = acc[k];
        }
}

template <int GROUP, int CHUNKS>
static hipError_t launch_t(const SpmmArgs &a, hipStream_t s) {
    constexpr int RPB = 4 * (64 / GROUP);
    const uint32_t nchunk = a.ld >> 2;
    dim3 grid((a.N + RPB - 1) / RPB, (nchunk + GROUP * CHUNKS - 1) / (GROUP * CHUNKS));
    if (a.N == 0 || nchunk == 0) return hipSuccess;
    hipLaunchKernelGGL((spmm_rows_kernel<GROUP, CHUNKS>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// variant 0 = auto.  slab = floats per feature slab (0 = whole row in one pass
// where it fits 4 chunks per lane).
hipError_t launch_spmm(const SpmmArgs &a, int variant, int slab, hipStream_t s) {
    (void)variant;
    if (a.ld & 3) return hipErrorInvalidValue;
    uint32_t width = a.ld;                  // floats handled by one block pass
    if (slab > 0 && (uint32_t)slab < width) width = (uint32_t)slab;
    const uint32_t ch = (width + 3) / 4;    // float4 chunks per row pass
    if (ch <= 8) return launch_t<8, 1>(a, s);
    if (ch <= 16) return launch_t<16, 1>(a, s);
    if (ch <= 32) return launch_t<32, 1>(a, s);
    if (ch <= 64) return launch_t<64, 1>(a, s);
    if (ch <= 128) return launch_t<64, 2>(a, s);
    if (ch <= 192) return launch_t<64, 3>(a, s);
    return launch_t<64, 4>(a, s);           // wider rows: gridDim.y slabs of 1024 floats
}

}  // namespace dory
