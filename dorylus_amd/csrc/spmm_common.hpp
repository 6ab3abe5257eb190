// spmm_common.hpp -- what the SpMM translation units share: lane-group broadcasts, the float4 multiply-add, and the three
// kernels that regroup an adjacency by source block (count -> scan -> stable fill), used by K1b's layout (spmm_blocked.hip:
// build_blocked) and by K1s's (spmm.hip: build_blocked_sweep).  Header-only (static kernels: one copy per translation unit).
#ifndef DORY_SPMM_COMMON_HPP
#define DORY_SPMM_COMMON_HPP
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

typedef float v4f __attribute__((ext_vector_type(4)));  // native vector for nontemporal ld/st

// `perm` (optional): position -> row (0xFFFFFFFF = empty position); `sblk` (optional): source row -> block.  Without
// them position = row and block = source / SB (K1b's layout); with them the K1s layout of build_blocked_sweep.
static __global__ __launch_bounds__(256) void blk_count_kernel(uint32_t N, const uint64_t *ptr, const uint32_t *idx,
                                                        uint32_t SB, uint32_t *cnt /*[nb][N]*/, const uint32_t *perm,
                                                        const uint16_t *sblk, const uint2 *slice) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const uint32_t v = perm ? perm[p] : p;
    if (v == 0xFFFFFFFFu) return;
    uint64_t e0 = ptr[v], e1 = ptr[v + 1];
    if (slice && slice[p].y > 1) {   // piece k of K of a split row: a contiguous range of its edge list
        const uint64_t chunk = (e1 - e0 + slice[p].y - 1) / slice[p].y;
        e0 = min(e1, e0 + (uint64_t)slice[p].x * chunk);
        e1 = min(e1, e0 + chunk);
    }
    for (uint64_t e = e0; e < e1; ++e) cnt[(size_t)(sblk ? (uint32_t)sblk[idx[e]] : idx[e] / SB) * N + p] += 1;
}

// one workgroup per block b: boff[b][0..N] = exclusive scan of cnt[b][0..N), total[b]
static __global__ __launch_bounds__(1024) void blk_scan_kernel(uint32_t N, const uint32_t *cnt, uint32_t *boff,
                                                        uint64_t *total) {
    __shared__ uint32_t part[1024];
    const uint32_t b = blockIdx.x, t = threadIdx.x;
    const uint32_t per = (N + 1023) / 1024;
    const uint32_t lo = min(N, t * per), hi = min(N, lo + per);
    const uint32_t *c = cnt + (size_t)b * N;
    uint32_t *o = boff + (size_t)b * (N + 1);
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += c[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 1024; ++i) {
            const uint32_t x = part[i];
            part[i] = run;
            run += x;
        }
        o[N] = run;
        total[b] = run;
    }
    __syncthreads();
    uint32_t run = part[t];
    for (uint32_t i = lo; i < hi; ++i) {
        o[i] = run;
        run += c[i];
    }
}

static __global__ __launch_bounds__(256) void blk_fill_kernel(uint32_t N, uint32_t nb, const uint64_t *ptr,
                                                       const uint32_t *idx, const float *val, uint32_t SB,
                                                       const uint64_t *bbase, const uint32_t *boff,
                                                       uint32_t *cursor /*[nb][N] scratch*/, uint32_t *bidx,
                                                       float *bval, const uint32_t *perm, const uint16_t *sblk,
                                                       const uint2 *slice, uint2 *bent = nullptr) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const uint32_t v = perm ? perm[p] : p;
    if (v == 0xFFFFFFFFu) return;
    for (uint32_t b = 0; b < nb; ++b) cursor[(size_t)b * N + p] = boff[(size_t)b * (N + 1) + p];
    uint64_t e0 = ptr[v], e1 = ptr[v + 1];
    if (slice && slice[p].y > 1) {
        const uint64_t chunk = (e1 - e0 + slice[p].y - 1) / slice[p].y;
        e0 = min(e1, e0 + (uint64_t)slice[p].x * chunk);
        e1 = min(e1, e0 + chunk);
    }
    for (uint64_t e = e0; e < e1; ++e) {   // original order inside each (block,row) segment
        const uint32_t s = idx[e];
        const uint32_t b = sblk ? (uint32_t)sblk[s] : s / SB;
        const uint64_t pos = bbase[b] + cursor[(size_t)b * N + p]++;
        if (bent) {
            bent[pos] = make_uint2(s, __float_as_uint(val[e]));
        } else {
            bidx[pos] = s;
            bval[pos] = val[e];
        }
    }
}

}  // namespace dory
#endif
