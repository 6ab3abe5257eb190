// spmm_lab.hip -- test bench for SpMM kernel candidates at Reddit scale (uniform graph), outside the C-ABI.
// Links the product library for build_blocked / K1 / K1b (reference + baseline).  GPU box only.
//   make -C tools/probes spmm_lab   (or see the hipcc line in tools/probes/Makefile)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../dorylus_amd/csrc/ctx.hpp"

using namespace dory;

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);       \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

__device__ __forceinline__ float4 fma4(float w, float4 x, float4 a) {
    a.x = fmaf(x.x, w, a.x);
    a.y = fmaf(x.y, w, a.y);
    a.z = fmaf(x.z, w, a.z);
    a.w = fmaf(x.w, w, a.w);
    return a;
}

// ---- K1s: destination rows split over the XCDs, register accumulators, every workgroup sweeps all source blocks ----
template <int GROUP, int R, int U, bool GH>
__global__ __launch_bounds__(256) void spmm_sweep_kernel(SpmmArgs a, BlockedAdj B, uint32_t rpx, uint32_t tiles_x,
                                                         uint32_t G, uint32_t *done /*[8][sweeps][nb] + flag*/, uint32_t nsweeps,
                                                         unsigned long long *dbg) {
    constexpr int GPW = 64 / GROUP;          // groups per wave
    constexpr int NGRP = 256 / GROUP;        // groups per workgroup
    constexpr int RW = NGRP * R;             // rows per workgroup
    constexpr int C = 128;                   // staged entries per group
    __shared__ uint2 stage[NGRP][C];
    __shared__ uint32_t lds_allowed, lds_cnt[8];
    if (threadIdx.x < 8) lds_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 8) lds_allowed = 0;
    __syncthreads();
    const uint32_t id = blockIdx.x, xcd = id & 7u, k = id >> 3;
    const uint32_t spp = G ? (tiles_x + G - 1) / G : 1;      // sweeps per slab
    const uint32_t tiles_pad = G ? spp * G : tiles_x;
    const uint32_t slab = k / tiles_pad, t = k % tiles_pad;
    if (t >= tiles_x) return;                                // padding workgroup of a slab's last sweep
    const uint32_t q = G ? k / G : 0;                        // sweep (global over slabs)
    const uint32_t cnt_q = G ? min(G, tiles_x - (q % spp) * G) : 0;
    const uint32_t cnt_p = G ? (q % spp == 0 ? min(G, tiles_x - (spp - 1) * G) : G) : 0;   // size of sweep q-1
    uint32_t *dq = done + ((size_t)xcd * nsweeps + q) * B.nb;
    uint32_t *nosync = done + (size_t)8 * nsweeps * B.nb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane % GROUP, gi = lane / GROUP;
    const int g = wave * GPW + gi;
    const uint32_t xend = min((xcd + 1) * rpx, a.N);
    const uint32_t v0 = min(xcd * rpx + t * RW + (uint32_t)g * R, xend);
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t col = slab * GROUP + li;
    const bool col_ok = col < nchunk;
    const uint32_t ccol = col_ok ? col : 0;
    const float4 *xl4 = reinterpret_cast<const float4 *>(a.xl);
    const float4 *xg4 = reinterpret_cast<const float4 *>(a.xg);
    uint2 *st = stage[g];

    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (uint32_t b = 0; b < B.nb; ++b) {
        unsigned long long t_a = 0;
        if (dbg && xcd == 0 && threadIdx.x == 0) t_a = wall_clock64();
        if (G) {   // start step b only when every workgroup of the sweep (or the previous one) has finished step b-2
            const uint32_t *w = b >= 2 ? dq + (b - 2) : (q > 0 ? dq - B.nb + (B.nb - 2 + b) : nullptr);
            const uint32_t need = b >= 2 ? cnt_q : cnt_p;
            if (w) {
                if (wave == 0) {      // one poller per workgroup; the other waves watch LDS
                    if (lane == 0) {
                        int spins = 0;
                        while (__hip_atomic_load(nosync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 &&
                               __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                            __builtin_amdgcn_s_sleep(32);
                            if (++spins > 3000) { __hip_atomic_fetch_add(nosync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        }
                        __hip_atomic_store(&lds_allowed, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
                    while (__hip_atomic_load(&lds_allowed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < b)
                        __builtin_amdgcn_s_sleep(8);
                }
            }
        }
        if (dbg && xcd == 0 && threadIdx.x == 0 && k < 2048 && b < 64) {
            dbg[((size_t)k * 64 + b) * 2] = t_a;
            dbg[((size_t)k * 64 + b) * 2 + 1] = wall_clock64();
        }
        const uint32_t *boff = B.boff + (size_t)b * (a.N + 1);
        const uint64_t base = B.bbase[b];
        const uint32_t my_o = boff[min(v0 + (uint32_t)min(li, R), xend)];
        uint32_t o[R + 1];
#pragma unroll
        for (int r = 0; r <= R; ++r) o[r] = (uint32_t)__shfl((int)my_o, r, GROUP);
        for (uint32_t cs = o[0]; cs < o[R]; cs += C) {
            const uint32_t ce = min(cs + C, o[R]);
            // stage entries [cs, ce) of this group
#pragma unroll
            for (int q = 0; q < C / GROUP; ++q) {
                const uint32_t p = cs + q * GROUP + li;
                if (p < ce) {
                    uint2 en;
                    en.x = __builtin_nontemporal_load(B.bidx + base + p);
                    en.y = __float_as_uint(__builtin_nontemporal_load(B.bval + base + p));
                    st[q * GROUP + li] = en;
                }
            }
            // same-wave LDS write -> read: the compiler inserts the lgkmcnt wait
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t lo = max(o[r], cs), hi = min(o[r + 1], ce);
                for (uint32_t e = lo; e < hi; e += U) {
                    float4 x[U];
                    float w[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const bool ok = e + u < hi;
                        const uint2 en = st[ok ? e + u - cs : 0];
                        const uint32_t s = en.x;
                        w[u] = ok ? __uint_as_float(en.y) : 0.f;
                        const float4 *row = (!GH || s < a.N) ? xl4 + (size_t)s * nchunk : xg4 + (size_t)(s - a.N) * nchunk;
                        x[u] = ok ? row[ccol] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) acc[r] = fma4(w[u], x[u], acc[r]);
                }
            }
        }
        if (G && lane == 0) {   // the last of the four waves to finish step b reports it
            const uint32_t old = __hip_atomic_fetch_add(&lds_cnt[b & 7], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == 3) {
                __hip_atomic_store(&lds_cnt[b & 7], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(dq + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    float4 *out4 = reinterpret_cast<float4 *>(a.out);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t v = v0 + r;
        if (v < xend && col_ok) {
            float4 o4 = acc[r];
            if (a.self_mode != 0) {
                const float sc = a.self_mode == 1 ? a.self_scale[v] : 1.f;
                o4 = fma4(sc, xl4[(size_t)v * nchunk + col], o4);
            }
            out4[(size_t)v * nchunk + col] = o4;
        }
    }
}

// ---- K1s v3: one 1024-thread workgroup per CU, cheap per-XCD gate (arrivals spread over a 128-B line), next window
// prefetched into L2 while the current one is used ----
constexpr int NT3 = 1024;
template <int GROUP, int R, int U, bool GH>
__global__ __launch_bounds__(NT3) void spmm_sweep3_kernel(SpmmArgs a, BlockedAdj B, uint32_t rpx, uint32_t tiles_x, uint32_t G,
                                                          uint32_t *done /*[8][sweeps][nb][32] + flag*/, uint32_t nsweeps,
                                                          int slack, int prefetch) {
    constexpr int GPW = 64 / GROUP;
    constexpr int NGRP = NT3 / GROUP;
    constexpr int NW = NT3 / 64;
    constexpr int RW = NGRP * R;
    constexpr int C = 128;
    __shared__ uint2 stage[NGRP][C];
    __shared__ uint32_t lds_allowed, lds_cnt[8];
    if (threadIdx.x < 8) lds_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 8) lds_allowed = 0;
    __syncthreads();
    const uint32_t id = blockIdx.x, xcd = id & 7u, k = id >> 3;
    const uint32_t spp = (tiles_x + G - 1) / G;
    const uint32_t tiles_pad = spp * G;
    const uint32_t slab = k / tiles_pad, t = k % tiles_pad;
    if (t >= tiles_x) return;
    const uint32_t q = k / G;
    const uint32_t cnt_q = min(G, tiles_x - (q % spp) * G);
    const uint32_t cnt_p = q % spp == 0 ? min(G, tiles_x - (spp - 1) * G) : G;
    uint32_t *dq = done + ((size_t)xcd * nsweeps + q) * B.nb * 32;
    uint32_t *nosync = done + (size_t)8 * nsweeps * B.nb * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane % GROUP, gi = lane / GROUP;
    const int g = wave * GPW + gi;
    const uint32_t xend = min((xcd + 1) * rpx, a.N);
    const uint32_t v0 = min(xcd * rpx + t * RW + (uint32_t)g * R, xend);
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t col = slab * GROUP + li;
    const bool col_ok = col < nchunk;
    const uint32_t ccol = col_ok ? col : 0;
    const float4 *xl4 = reinterpret_cast<const float4 *>(a.xl);
    const float4 *xg4 = reinterpret_cast<const float4 *>(a.xg);
    uint2 *st = stage[g];
    const uint32_t NGv = B.nb * B.SB;   // >= rows of the virtual source space (lab: no ghosts)

    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    float pf_sink = 0.f;

    for (uint32_t b = 0; b < B.nb; ++b) {
        // gate: start step b only when every workgroup has finished step b-slack-1 (of this sweep or the previous one)
        {
            const int bb = (int)b - slack - 1;
            const uint32_t *w = bb >= 0 ? dq + (size_t)bb * 32 : (q > 0 ? dq - (size_t)B.nb * 32 + (size_t)((int)B.nb + bb) * 32 : nullptr);
            const uint32_t need = bb >= 0 ? cnt_q : cnt_p;
            if (w) {
                if (wave == 0) {
                    if (lane == 0) {
                        int spins = 0;
                        while (true) {
                            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                            uint32_t sum = 0;
#pragma unroll
                            for (int i = 0; i < 32; i += 4) {
                                u4 v;
                                const uint32_t *p = w + i;
                                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
                                sum += v.x + v.y + v.z + v.w;
                            }
                            if (sum >= need) break;
                            if (__hip_atomic_load(nosync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                            __builtin_amdgcn_s_sleep(4);
                            if (++spins > 20000) { __hip_atomic_fetch_add(nosync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        }
                        __hip_atomic_store(&lds_allowed, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
                    while (__hip_atomic_load(&lds_allowed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < b)
                        __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        float pf[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (prefetch && b + 1 < B.nb) {   // this workgroup's share of window b+1 (consumed at the end of the step)
            const uint32_t tq = t % G;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const uint32_t i = (tq + (uint32_t)j * cnt_q) * (NW * GPW) + (uint32_t)(wave * GPW + gi);
                const uint32_t srow = (b + 1) * B.SB + i;
                if (i < B.SB && srow < a.N && col_ok)
                    pf[j] = __builtin_nontemporal_load(reinterpret_cast<const float *>(xl4 + (size_t)srow * nchunk + ccol));
            }
        }
        const uint32_t *boff = B.boff + (size_t)b * (a.N + 1);
        const uint64_t base = B.bbase[b];
        const uint32_t my_o = boff[min(v0 + (uint32_t)min(li, R), xend)];
        uint32_t o[R + 1];
#pragma unroll
        for (int r = 0; r <= R; ++r) o[r] = (uint32_t)__shfl((int)my_o, r, GROUP);
        for (uint32_t cs = o[0]; cs < o[R]; cs += C) {
            const uint32_t ce = min(cs + C, o[R]);
#pragma unroll
            for (int qq = 0; qq < C / GROUP; ++qq) {
                const uint32_t p = cs + qq * GROUP + li;
                if (p < ce) {
                    uint2 en;
                    en.x = __builtin_nontemporal_load(B.bidx + base + p);
                    en.y = __float_as_uint(__builtin_nontemporal_load(B.bval + base + p));
                    st[qq * GROUP + li] = en;
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t lo = max(o[r], cs), hi = min(o[r + 1], ce);
                for (uint32_t e = lo; e < hi; e += U) {
                    float4 x[U];
                    float w[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const bool ok = e + u < hi;
                        const uint2 en = st[ok ? e + u - cs : 0];
                        const uint32_t s = en.x;
                        w[u] = ok ? __uint_as_float(en.y) : 0.f;
                        const float4 *row = (!GH || s < a.N) ? xl4 + (size_t)s * nchunk : xg4 + (size_t)(s - a.N) * nchunk;
                        x[u] = ok ? row[ccol] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) acc[r] = fma4(w[u], x[u], acc[r]);
                }
            }
        }
        pf_sink += (pf[0] + pf[1]) + (pf[2] + pf[3]) + (pf[4] + pf[5]);
        if (lane == 0) {   // the last wave to finish step b reports it
            const uint32_t old = __hip_atomic_fetch_add(&lds_cnt[b & 7], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == NW - 1) {
                __hip_atomic_store(&lds_cnt[b & 7], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(dq + (size_t)b * 32 + (t & 31), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    (void)NGv;
    if (pf_sink == 1.2345e-30f) a.out[0] = pf_sink;   // keeps the prefetch loads alive
    float4 *out4 = reinterpret_cast<float4 *>(a.out);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t v = v0 + r;
        if (v < xend && col_ok) {
            float4 o4 = acc[r];
            if (a.self_mode != 0) {
                const float sc = a.self_mode == 1 ? a.self_scale[v] : 1.f;
                o4 = fma4(sc, xl4[(size_t)v * nchunk + col], o4);
            }
            out4[(size_t)v * nchunk + col] = o4;
        }
    }
}

// ---- K1s v4 (v3 + next step's offsets/entries prefetched, window prefetch by line): one 1024-thread workgroup per CU, cheap per-XCD gate (arrivals spread over a 128-B line), next window
// prefetched into L2 while the current one is used ----

template <int GROUP, int R, int U, bool GH>
__global__ __launch_bounds__(NT3) void spmm_sweep4_kernel(SpmmArgs a, BlockedAdj B, uint32_t rpx, uint32_t tiles_x, uint32_t G,
                                                          uint32_t *done /*[8][sweeps][nb][32] + flag*/, uint32_t nsweeps,
                                                          int slack, int prefetch, int lag_w, int lag_g) {
    constexpr int GPW = 64 / GROUP;
    constexpr int NGRP = NT3 / GROUP;
    constexpr int NW = NT3 / 64;
    constexpr int RW = NGRP * R;
    constexpr int C = 128;
    __shared__ uint2 stage[NGRP][C];
    __shared__ uint32_t lds_allowed, lds_cnt[8];
    if (threadIdx.x < 8) lds_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 8) lds_allowed = 0;
    __syncthreads();
    const uint32_t id = blockIdx.x, xcd = id & 7u, k = id >> 3;
    const uint32_t spp = (tiles_x + G - 1) / G;
    const uint32_t tiles_pad = spp * G;
    const uint32_t slab = k / tiles_pad, t = k % tiles_pad;
    if (t >= tiles_x) return;
    const uint32_t q = k / G;
    const uint32_t cnt_q = min(G, tiles_x - (q % spp) * G);
    const uint32_t cnt_p = q % spp == 0 ? min(G, tiles_x - (spp - 1) * G) : G;
    uint32_t *dq = done + ((size_t)xcd * nsweeps + q) * B.nb * 32;
    uint32_t *nosync = done + (size_t)8 * nsweeps * B.nb * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane % GROUP, gi = lane / GROUP;
    const int g = wave * GPW + gi;
    const uint32_t xend = min((xcd + 1) * rpx, a.N);
    const uint32_t v0 = min(xcd * rpx + t * RW + (uint32_t)g * R, xend);
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t col = slab * GROUP + li;
    const bool col_ok = col < nchunk;
    const uint32_t ccol = col_ok ? col : 0;
    const float4 *xl4 = reinterpret_cast<const float4 *>(a.xl);
    const float4 *xg4 = reinterpret_cast<const float4 *>(a.xg);
    uint2 *st = stage[g];
    const uint32_t NGv = B.nb * B.SB;   // >= rows of the virtual source space (lab: no ghosts)

    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    float pf_sink = 0.f;
    uint32_t my_o = B.boff[min(v0 + (uint32_t)min(li, R), xend)];
    uint2 en_pre[C / GROUP];
    {
        const uint64_t base0 = B.bbase[0];
        const uint32_t o0 = (uint32_t)__shfl((int)my_o, 0, GROUP), oR = (uint32_t)__shfl((int)my_o, R, GROUP);
#pragma unroll
        for (int qq = 0; qq < C / GROUP; ++qq) {
            const uint32_t p = o0 + qq * GROUP + li;
            en_pre[qq] = make_uint2(0u, 0u);
            if (p < oR) {
                en_pre[qq].x = __builtin_nontemporal_load(B.bidx + base0 + p);
                en_pre[qq].y = __float_as_uint(__builtin_nontemporal_load(B.bval + base0 + p));
            }
        }
    }

    for (uint32_t b = 0; b < B.nb; ++b) {
        // gate: start step b only when every workgroup has finished step b-slack-1 (of this sweep or the previous one)
        {
            const int bb = (int)b - slack - 1;
            const uint32_t *w = bb >= 0 ? dq + (size_t)bb * 32 : (q > 0 ? dq - (size_t)B.nb * 32 + (size_t)((int)B.nb + bb) * 32 : nullptr);
            const uint32_t need0 = bb >= 0 ? cnt_q : cnt_p;
            const uint32_t need = need0 > (uint32_t)lag_g ? need0 - (uint32_t)lag_g : 1u;
            if (w) {
                if (wave == 0) {
                    if (lane == 0) {
                        int spins = 0;
                        while (true) {
                            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                            uint32_t sum = 0;
#pragma unroll
                            for (int i = 0; i < 32; i += 4) {
                                u4 v;
                                const uint32_t *p = w + i;
                                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
                                sum += v.x + v.y + v.z + v.w;
                            }
                            if (sum >= need) break;
                            if (__hip_atomic_load(nosync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                            __builtin_amdgcn_s_sleep(4);
                            if (++spins > 20000) { __hip_atomic_fetch_add(nosync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        }
                        __hip_atomic_store(&lds_allowed, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
                    while (__hip_atomic_load(&lds_allowed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < b)
                        __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        // (1) offsets of step b+1 and this workgroup's share of window b+1 (one 128-B line per lane)
        uint32_t my_o_next = 0;
        if (b + 1 < B.nb) my_o_next = (B.boff + (size_t)(b + 1) * (a.N + 1))[min(v0 + (uint32_t)min(li, R), xend)];
        float pf[2] = {0.f, 0.f};
        if (prefetch && b + 1 < B.nb) {
            const uint32_t L = B.SB * (GROUP / 8);          // 128-B lines of the window (this slab)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t l = ((t % G) + (uint32_t)j * cnt_q) * NT3 + threadIdx.x;
                const uint32_t srow = (b + 1) * B.SB + l / (GROUP / 8);
                const uint32_t c4 = slab * GROUP + (l % (GROUP / 8)) * 8;
                if (l < L && srow < a.N && c4 < nchunk) pf[j] = reinterpret_cast<const float *>(xl4 + (size_t)srow * nchunk + c4)[0];
            }
        }
        // (2) this step: entries of the first chunk were loaded at the end of the previous step
        const uint64_t base = B.bbase[b];
        uint32_t o[R + 1];
#pragma unroll
        for (int r = 0; r <= R; ++r) o[r] = (uint32_t)__shfl((int)my_o, r, GROUP);
        for (uint32_t cs = o[0]; cs < o[R]; cs += C) {
            const uint32_t ce = min(cs + C, o[R]);
            if (cs == o[0]) {
#pragma unroll
                for (int qq = 0; qq < C / GROUP; ++qq)
                    if (cs + qq * GROUP + li < ce) st[qq * GROUP + li] = en_pre[qq];
            } else {
#pragma unroll
                for (int qq = 0; qq < C / GROUP; ++qq) {
                    const uint32_t p = cs + qq * GROUP + li;
                    if (p < ce) {
                        uint2 en;
                        en.x = __builtin_nontemporal_load(B.bidx + base + p);
                        en.y = __float_as_uint(__builtin_nontemporal_load(B.bval + base + p));
                        st[qq * GROUP + li] = en;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t lo = max(o[r], cs), hi = min(o[r + 1], ce);
                for (uint32_t e = lo; e < hi; e += U) {
                    float4 x[U];
                    float w[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const bool ok = e + u < hi;
                        const uint2 en = st[ok ? e + u - cs : 0];
                        const uint32_t s = en.x;
                        w[u] = ok ? __uint_as_float(en.y) : 0.f;
                        const float4 *row = (!GH || s < a.N) ? xl4 + (size_t)s * nchunk : xg4 + (size_t)(s - a.N) * nchunk;
                        x[u] = ok ? row[ccol] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) acc[r] = fma4(w[u], x[u], acc[r]);
                }
            }
        }
        // (3) first chunk of step b+1's entries (in flight across the gate)
        my_o = my_o_next;
        if (b + 1 < B.nb) {
            const uint64_t base1 = B.bbase[b + 1];
            const uint32_t o0 = (uint32_t)__shfl((int)my_o, 0, GROUP), oR = (uint32_t)__shfl((int)my_o, R, GROUP);
#pragma unroll
            for (int qq = 0; qq < C / GROUP; ++qq) {
                const uint32_t p = o0 + qq * GROUP + li;
                if (p < oR) {
                    en_pre[qq].x = __builtin_nontemporal_load(B.bidx + base1 + p);
                    en_pre[qq].y = __float_as_uint(__builtin_nontemporal_load(B.bval + base1 + p));
                }
            }
        }
        pf_sink += pf[0] + pf[1];
        if (lane == 0) {   // the last wave to finish step b reports it
            const uint32_t old = __hip_atomic_fetch_add(&lds_cnt[b & 7], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == (uint32_t)(NW - 1 - lag_w))
                __hip_atomic_fetch_add(dq + (size_t)b * 32 + (t & 31), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == NW - 1) __hip_atomic_store(&lds_cnt[b & 7], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    (void)NGv;
    if (pf_sink == 1.2345e-30f) a.out[0] = pf_sink;   // keeps the prefetch loads alive
    float4 *out4 = reinterpret_cast<float4 *>(a.out);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t v = v0 + r;
        if (v < xend && col_ok) {
            float4 o4 = acc[r];
            if (a.self_mode != 0) {
                const float sc = a.self_mode == 1 ? a.self_scale[v] : 1.f;
                o4 = fma4(sc, xl4[(size_t)v * nchunk + col], o4);
            }
            out4[(size_t)v * nchunk + col] = o4;
        }
    }
}

// ---- K1s v5 (FORM 0: full batches + tail, FORM 1: row pairs) -- from v4 (v3 + next step's offsets/entries prefetched, window prefetch by line): one 1024-thread workgroup per CU, cheap per-XCD gate (arrivals spread over a 128-B line), next window
// prefetched into L2 while the current one is used ----

template <int GROUP, int R, int U, bool GH, int FORM>
__global__ __launch_bounds__(NT3) void spmm_sweep5_kernel(SpmmArgs a, BlockedAdj B, uint32_t rpx, uint32_t tiles_x, uint32_t G,
                                                          uint32_t *done /*[8][sweeps][nb][32] + flag*/, uint32_t nsweeps,
                                                          int slack, int prefetch, int lag_w, int lag_g, unsigned long long *tacc) {
    constexpr int GPW = 64 / GROUP;
    constexpr int NGRP = NT3 / GROUP;
    constexpr int NW = NT3 / 64;
    constexpr int RW = NGRP * R;
    constexpr int C = 128;
    __shared__ uint2 stage[NGRP][C];
    __shared__ uint32_t lds_allowed, lds_cnt[8];
    if (threadIdx.x < 8) lds_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 8) lds_allowed = 0;
    __syncthreads();
    const uint32_t id = blockIdx.x, xcd = id & 7u, k = id >> 3;
    const uint32_t spp = (tiles_x + G - 1) / G;
    const uint32_t tiles_pad = spp * G;
    const uint32_t slab = k / tiles_pad, t = k % tiles_pad;
    if (t >= tiles_x) return;
    const uint32_t q = k / G;
    const uint32_t cnt_q = min(G, tiles_x - (q % spp) * G);
    const uint32_t cnt_p = q % spp == 0 ? min(G, tiles_x - (spp - 1) * G) : G;
    uint32_t *dq = done + ((size_t)xcd * nsweeps + q) * B.nb * 32;
    uint32_t *nosync = done + (size_t)8 * nsweeps * B.nb * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane % GROUP, gi = lane / GROUP;
    const int g = wave * GPW + gi;
    const uint32_t xend = min((xcd + 1) * rpx, a.N);
    const uint32_t v0 = min(xcd * rpx + t * RW + (uint32_t)g * R, xend);
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t col = slab * GROUP + li;
    const bool col_ok = col < nchunk;
    const uint32_t ccol = col_ok ? col : 0;
    const float4 *xl4 = reinterpret_cast<const float4 *>(a.xl);
    const float4 *xg4 = reinterpret_cast<const float4 *>(a.xg);
    uint2 *st = stage[g];
    const uint32_t NGv = B.nb * B.SB;   // >= rows of the virtual source space (lab: no ghosts)

    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    float pf_sink = 0.f;
    uint32_t my_o = B.boff[min(v0 + (uint32_t)min(li, R), xend)];
    uint2 en_pre[C / GROUP];
    {
        const uint64_t base0 = B.bbase[0];
        const uint32_t o0 = (uint32_t)__shfl((int)my_o, 0, GROUP), oR = (uint32_t)__shfl((int)my_o, R, GROUP);
#pragma unroll
        for (int qq = 0; qq < C / GROUP; ++qq) {
            const uint32_t p = o0 + qq * GROUP + li;
            en_pre[qq] = make_uint2(0u, 0u);
            if (p < oR) {
                en_pre[qq].x = __builtin_nontemporal_load(B.bidx + base0 + p);
                en_pre[qq].y = __float_as_uint(__builtin_nontemporal_load(B.bval + base0 + p));
            }
        }
    }

    unsigned long long t_gate = 0, t_stage = 0, t_rows = 0, t_post = 0;
    for (uint32_t b = 0; b < B.nb; ++b) {
        unsigned long long tq0 = wall_clock64();
        // gate: start step b only when every workgroup has finished step b-slack-1 (of this sweep or the previous one)
        {
            const int bb = (int)b - slack - 1;
            const uint32_t *w = bb >= 0 ? dq + (size_t)bb * 32 : (q > 0 ? dq - (size_t)B.nb * 32 + (size_t)((int)B.nb + bb) * 32 : nullptr);
            const uint32_t need0 = bb >= 0 ? cnt_q : cnt_p;
            const uint32_t need = need0 > (uint32_t)lag_g ? need0 - (uint32_t)lag_g : 1u;
            if (w) {
                if (wave == 0) {
                    if (lane == 0) {
                        int spins = 0;
                        while (true) {
                            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                            uint32_t sum = 0;
#pragma unroll
                            for (int i = 0; i < 32; i += 4) {
                                u4 v;
                                const uint32_t *p = w + i;
                                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
                                sum += v.x + v.y + v.z + v.w;
                            }
                            if (sum >= need) break;
                            if (__hip_atomic_load(nosync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                            __builtin_amdgcn_s_sleep(4);
                            if (++spins > 20000) { __hip_atomic_fetch_add(nosync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                        }
                        __hip_atomic_store(&lds_allowed, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
                    while (__hip_atomic_load(&lds_allowed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < b)
                        __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        unsigned long long tq1 = wall_clock64();
        t_gate += tq1 - tq0;
        // (1) offsets of step b+1 and this workgroup's share of window b+1 (one 128-B line per lane)
        uint32_t my_o_next = 0;
        if (b + 1 < B.nb) my_o_next = (B.boff + (size_t)(b + 1) * (a.N + 1))[min(v0 + (uint32_t)min(li, R), xend)];
        float pf[2] = {0.f, 0.f};
        if (prefetch && b + 1 < B.nb) {
            const uint32_t L = B.SB * (GROUP / 8);          // 128-B lines of the window (this slab)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t l = ((t % G) + (uint32_t)j * cnt_q) * NT3 + threadIdx.x;
                const uint32_t srow = (b + 1) * B.SB + l / (GROUP / 8);
                const uint32_t c4 = slab * GROUP + (l % (GROUP / 8)) * 8;
                if (l < L && srow < a.N && c4 < nchunk) pf[j] = reinterpret_cast<const float *>(xl4 + (size_t)srow * nchunk + c4)[0];
            }
        }
        // (2) this step: entries of the first chunk were loaded at the end of the previous step
        const uint64_t base = B.bbase[b];
        uint32_t o[R + 1];
#pragma unroll
        for (int r = 0; r <= R; ++r) o[r] = (uint32_t)__shfl((int)my_o, r, GROUP);
        for (uint32_t cs = o[0]; cs < o[R]; cs += C) {
            const uint32_t ce = min(cs + C, o[R]);
            if (cs == o[0]) {
#pragma unroll
                for (int qq = 0; qq < C / GROUP; ++qq)
                    if (cs + qq * GROUP + li < ce) st[qq * GROUP + li] = en_pre[qq];
            } else {
#pragma unroll
                for (int qq = 0; qq < C / GROUP; ++qq) {
                    const uint32_t p = cs + qq * GROUP + li;
                    if (p < ce) {
                        uint2 en;
                        en.x = __builtin_nontemporal_load(B.bidx + base + p);
                        en.y = __float_as_uint(__builtin_nontemporal_load(B.bval + base + p));
                        st[qq * GROUP + li] = en;
                    }
                }
            }
            unsigned long long tq2 = wall_clock64();
            t_stage += tq2 - tq1;
            auto rowp = [&](uint32_t sidx) -> const float4 * {
                return ((!GH || sidx < a.N) ? xl4 + (size_t)sidx * nchunk : xg4 + (size_t)(sidx - a.N) * nchunk) + ccol;
            };
            if constexpr (FORM == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t lo = max(o[r], cs), hi = min(o[r + 1], ce);
                    uint32_t e = lo;
                    for (; e + U <= hi; e += U) {          // full batches: nothing predicated
                        uint2 en[U];
                        float4 x[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) en[u] = st[e + u - cs];
#pragma unroll
                        for (int u = 0; u < U; ++u) x[u] = *rowp(en[u].x);
#pragma unroll
                        for (int u = 0; u < U; ++u) acc[r] = fma4(__uint_as_float(en[u].y), x[u], acc[r]);
                    }
                    if (e < hi) {                           // tail: 1 .. U-1 edges
                        const uint32_t n = hi - e;
                        uint2 en[U - 1];
                        float4 x[U - 1];
#pragma unroll
                        for (int u = 0; u < U - 1; ++u) en[u] = st[min(e + u, hi - 1) - cs];
#pragma unroll
                        for (int u = 0; u < U - 1; ++u) x[u] = (uint32_t)u < n ? *rowp(en[u].x) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int u = 0; u < U - 1; ++u) acc[r] = fma4((uint32_t)u < n ? __uint_as_float(en[u].y) : 0.f, x[u], acc[r]);
                    }
                }
            } else {                                        // two rows at a time: 2U gathers in flight per group
#pragma unroll
                for (int r = 0; r < R; r += 2) {
                    const uint32_t lo0 = max(o[r], cs), hi0 = min(o[r + 1], ce);
                    const uint32_t lo1 = r + 1 < R ? max(o[r + 1], cs) : 0u, hi1 = r + 1 < R ? min(o[r + 2 <= R ? r + 2 : R], ce) : 0u;
                    uint32_t e0 = lo0, e1 = lo1;
                    while (e0 < hi0 || e1 < hi1) {
                        uint2 en0[U], en1[U];
                        float4 x0[U], x1[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            en0[u] = st[min(e0 + u, max(hi0, lo0 + 1) - 1) - cs];
                            en1[u] = st[min(e1 + u, max(hi1, lo1 + 1) - 1) - cs];
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            x0[u] = e0 + u < hi0 ? *rowp(en0[u].x) : make_float4(0.f, 0.f, 0.f, 0.f);
                            x1[u] = e1 + u < hi1 ? *rowp(en1[u].x) : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            acc[r] = fma4(e0 + u < hi0 ? __uint_as_float(en0[u].y) : 0.f, x0[u], acc[r]);
                            if (r + 1 < R) acc[r + 1 < R ? r + 1 : r] = fma4(e1 + u < hi1 ? __uint_as_float(en1[u].y) : 0.f, x1[u], acc[r + 1 < R ? r + 1 : r]);
                        }
                        e0 = min(e0 + U, hi0 > e0 ? hi0 : e0);
                        e1 = min(e1 + U, hi1 > e1 ? hi1 : e1);
                        if (e0 + 0 >= hi0) e0 = hi0 > e0 ? hi0 : e0;
                        if (e1 + 0 >= hi1) e1 = hi1 > e1 ? hi1 : e1;
                    }
                }
            }
        }
        unsigned long long tq3 = wall_clock64();
        // (3) first chunk of step b+1's entries (in flight across the gate)
        my_o = my_o_next;
        if (b + 1 < B.nb) {
            const uint64_t base1 = B.bbase[b + 1];
            const uint32_t o0 = (uint32_t)__shfl((int)my_o, 0, GROUP), oR = (uint32_t)__shfl((int)my_o, R, GROUP);
#pragma unroll
            for (int qq = 0; qq < C / GROUP; ++qq) {
                const uint32_t p = o0 + qq * GROUP + li;
                if (p < oR) {
                    en_pre[qq].x = __builtin_nontemporal_load(B.bidx + base1 + p);
                    en_pre[qq].y = __float_as_uint(__builtin_nontemporal_load(B.bval + base1 + p));
                }
            }
        }
        pf_sink += pf[0] + pf[1];
        { unsigned long long tq4 = wall_clock64(); t_post += tq4 - tq3; t_rows += tq3 - tq1; }
        if (lane == 0) {   // the last wave to finish step b reports it
            const uint32_t old = __hip_atomic_fetch_add(&lds_cnt[b & 7], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == (uint32_t)(NW - 1 - lag_w))
                __hip_atomic_fetch_add(dq + (size_t)b * 32 + (t & 31), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == NW - 1) __hip_atomic_store(&lds_cnt[b & 7], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (tacc && lane == 0) { atomicAdd(tacc + 0, t_gate); atomicAdd(tacc + 1, t_stage); atomicAdd(tacc + 2, t_rows); atomicAdd(tacc + 3, t_post); atomicAdd(tacc + 4, 1ull); }
    (void)NGv;
    if (pf_sink == 1.2345e-30f) a.out[0] = pf_sink;   // keeps the prefetch loads alive
    float4 *out4 = reinterpret_cast<float4 *>(a.out);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t v = v0 + r;
        if (v < xend && col_ok) {
            float4 o4 = acc[r];
            if (a.self_mode != 0) {
                const float sc = a.self_mode == 1 ? a.self_scale[v] : 1.f;
                o4 = fma4(sc, xl4[(size_t)v * nchunk + col], o4);
            }
            out4[(size_t)v * nchunk + col] = o4;
        }
    }
}

template <int GROUP, int R, int U>
static void launch_sweep3(const SpmmArgs &a, const BlockedAdj &B, uint32_t *done, int slack, int prefetch, hipStream_t s) {
    constexpr int RW = (NT3 / GROUP) * R;
    const uint32_t G = 32;
    const uint32_t rpx = (a.N + 7) / 8;
    const uint32_t tiles_x = (rpx + RW - 1) / RW;
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t slabs = (nchunk + GROUP - 1) / GROUP;
    const uint32_t spp = (tiles_x + G - 1) / G, tiles_pad = spp * G;
    const uint32_t nsweeps = slabs * spp;
    CK(hipMemsetAsync(done, 0, ((size_t)8 * nsweeps * B.nb * 32 + 1) * 4, s));
    hipLaunchKernelGGL((spmm_sweep3_kernel<GROUP, R, U, false>), dim3(8 * slabs * tiles_pad), dim3(NT3), 0, s, a, B, rpx,
                       tiles_x, G, done, nsweeps, slack, prefetch);
}

template <int GROUP, int R, int U>
static void launch_sweep4(const SpmmArgs &a, const BlockedAdj &B, uint32_t *done, int slack, int prefetch, hipStream_t s,
                          int lag_w = 0, int lag_g = 0) {
    constexpr int RW = (NT3 / GROUP) * R;
    const uint32_t G = 32;
    const uint32_t rpx = (a.N + 7) / 8;
    const uint32_t tiles_x = (rpx + RW - 1) / RW;
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t slabs = (nchunk + GROUP - 1) / GROUP;
    const uint32_t spp = (tiles_x + G - 1) / G, tiles_pad = spp * G;
    const uint32_t nsweeps = slabs * spp;
    CK(hipMemsetAsync(done, 0, ((size_t)8 * nsweeps * B.nb * 32 + 1) * 4, s));
    hipLaunchKernelGGL((spmm_sweep4_kernel<GROUP, R, U, false>), dim3(8 * slabs * tiles_pad), dim3(NT3), 0, s, a, B, rpx,
                       tiles_x, G, done, nsweeps, slack, prefetch, lag_w, lag_g);
}

static unsigned long long *g_tacc = nullptr;
template <int GROUP, int R, int U, int FORM>
static void launch_sweep5(const SpmmArgs &a, const BlockedAdj &B, uint32_t *done, int slack, int prefetch, hipStream_t s,
                          int lag_w = 0, int lag_g = 0) {
    constexpr int RW = (NT3 / GROUP) * R;
    const uint32_t G = 32;
    const uint32_t rpx = (a.N + 7) / 8;
    const uint32_t tiles_x = (rpx + RW - 1) / RW;
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t slabs = (nchunk + GROUP - 1) / GROUP;
    const uint32_t spp = (tiles_x + G - 1) / G, tiles_pad = spp * G;
    const uint32_t nsweeps = slabs * spp;
    CK(hipMemsetAsync(done, 0, ((size_t)8 * nsweeps * B.nb * 32 + 1) * 4, s));
    hipLaunchKernelGGL((spmm_sweep5_kernel<GROUP, R, U, false, FORM>), dim3(8 * slabs * tiles_pad), dim3(NT3), 0, s, a, B, rpx,
                       tiles_x, G, done, nsweeps, slack, prefetch, lag_w, lag_g, g_tacc);
}

template <int GROUP, int R, int U>
static void launch_sweep(const SpmmArgs &a, const BlockedAdj &B, uint32_t G, uint32_t *done, hipStream_t s,
                         unsigned long long *dbg = nullptr) {
    constexpr int RW = (256 / GROUP) * R;
    const uint32_t rpx = (a.N + 7) / 8;
    const uint32_t tiles_x = (rpx + RW - 1) / RW;
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t slabs = (nchunk + GROUP - 1) / GROUP;
    const uint32_t spp = G ? (tiles_x + G - 1) / G : 1, tiles_pad = G ? spp * G : tiles_x;
    const uint32_t nsweeps = slabs * spp;
    if (G) CK(hipMemsetAsync(done, 0, ((size_t)8 * nsweeps * B.nb + 1) * 4, s));
    hipLaunchKernelGGL((spmm_sweep_kernel<GROUP, R, U, false>), dim3(8 * slabs * tiles_pad), dim3(256), 0, s, a, B, rpx,
                       tiles_x, G, done, nsweeps, dbg);
}
template <int GROUP, int R, int U>
static int sweep_occupancy() {
    int nblk = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, spmm_sweep_kernel<GROUP, R, U, false>, 256, 0));
    return nblk;
}

static double rel_err(const std::vector<float> &a, const std::vector<float> &b) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        num = std::max(num, (double)std::fabs(a[i] - b[i]));
        den = std::max(den, (double)std::fabs(b[i]));
    }
    return num / (den + 1e-30);
}

template <class F>
static float time_ms(F f, int iters = 3) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char **argv) {
    const uint32_t N = 232965;
    const uint32_t deg = argc > 2 ? atoi(argv[2]) : 492;
    const uint32_t F = argc > 1 ? atoi(argv[1]) : 602;
    const uint32_t ld = (F + 31) & ~31u;
    const uint64_t E = (uint64_t)N * deg;
    const uint32_t win = argc > 3 ? atoi(argv[3]) : N;   // sources drawn from [0, win): L2-resident gather probe
    printf("N=%u deg=%u E=%llu F=%u ld=%u win=%u\n", N, deg, (unsigned long long)E, F, ld, win);
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
    // reference: K1 row kernel
    float t_ref = time_ms([&] { CK(launch_spmm(a, 0, 64, 0)); }, 1);
    printf("K1 (slab 64): %.3f ms\n", t_ref);
    std::vector<float> href((size_t)N * ld), hout((size_t)N * ld);
    CK(hipMemcpy(href.data(), d_ref, href.size() * 4, hipMemcpyDeviceToHost));
    a.out = d_out;
    const double gather = (double)E * ld * 4;

    // baseline K1b
    {
        BlockedAdj B{};
        CK(build_blocked(d_ptr, d_idx, d_val, N, N, E, 0, 512, &B, 0));
        float *partial;
        CK(hipMalloc(&partial, (size_t)B.nb * N * ld * 4));
        float t = time_ms([&] { CK(launch_spmm_blocked(a, B, partial, 32, nullptr, 0)); });
        CK(hipMemcpy(hout.data(), d_out, hout.size() * 4, hipMemcpyDeviceToHost));
        printf("K1b nb=%u group 32: %.3f ms  gather %.2f TB/s  err %.2e\n", B.nb, t, gather / t / 1e9, rel_err(hout, href));
        CK(hipFree(partial));
        free_blocked(&B);
    }
    uint32_t *d_done;
    CK(hipMalloc(&d_done, 64 << 20));
    CK(hipMalloc(&g_tacc, 64));
    for (uint32_t nb : {48u}) {
        BlockedAdj B{};
        CK(build_blocked(d_ptr, d_idx, d_val, N, N, E, nb, 512, &B, 0));
#define RUN(GR, R, U, SYNC)                                                                                       \
    do {                                                                                                          \
        CK(hipMemset(d_out, 0, (size_t)N * ld * 4));                                                              \
        const int occ = sweep_occupancy<GR, R, U>();                                                              \
        const uint32_t Gs = (SYNC) ? 32u * occ : 0u;                                                              \
        float t = time_ms([&] { launch_sweep<GR, R, U>(a, B, Gs, d_done, 0); });                                  \
        CK(hipMemcpy(hout.data(), d_out, hout.size() * 4, hipMemcpyDeviceToHost));                                \
        uint32_t flag = 0;                                                                                        \
        if (Gs) { const uint32_t RWm = (256 / GR) * R, rpxm = (N + 7) / 8, tx = (rpxm + RWm - 1) / RWm;            \
                  const uint32_t nsw = ((ld / 4 + GR - 1) / GR) * ((tx + Gs - 1) / Gs);                            \
                  CK(hipMemcpy(&flag, d_done + (size_t)8 * nsw * B.nb, 4, hipMemcpyDeviceToHost)); }               \
        printf("K1s nb=%3u window %.2f MB group %d R=%d U=%d occ %d sync %u timeouts %u: %.3f ms  gather %.2f TB/s  err %.2e\n", \
               B.nb, (double)B.SB * GR * 16 / 1048576.0, GR, R, U, occ, Gs, flag, t, gather / t / 1e9, rel_err(hout, href)); \
        (void)flag;                                                                                               \
        fflush(stdout);                                                                                           \
    } while (0)
#define RUN3(GR, R, U, SLACK, PF)                                                                                 \
    do {                                                                                                          \
        CK(hipMemset(d_out, 0, (size_t)N * ld * 4));                                                              \
        float t = time_ms([&] { launch_sweep3<GR, R, U>(a, B, d_done, SLACK, PF, 0); });                          \
        CK(hipMemcpy(hout.data(), d_out, hout.size() * 4, hipMemcpyDeviceToHost));                                \
        printf("K1s3 nb=%3u window %.2f MB group %d R=%d U=%d slack %d prefetch %d: %.3f ms  gather %.2f TB/s  err %.2e\n", \
               B.nb, (double)B.SB * GR * 16 / 1048576.0, GR, R, U, SLACK, PF, t, gather / t / 1e9, rel_err(hout, href)); \
        fflush(stdout);                                                                                           \
    } while (0)
#define RUN4(GR, R, U, SLACK, PF)                                                                                 \
    do {                                                                                                          \
        CK(hipMemset(d_out, 0, (size_t)N * ld * 4));                                                              \
        float t = time_ms([&] { launch_sweep4<GR, R, U>(a, B, d_done, SLACK, PF, 0); });                          \
        CK(hipMemcpy(hout.data(), d_out, hout.size() * 4, hipMemcpyDeviceToHost));                                \
        printf("K1s4 nb=%3u window %.2f MB group %d R=%d U=%d slack %d prefetch %d: %.3f ms  gather %.2f TB/s  err %.2e\n", \
               B.nb, (double)B.SB * GR * 16 / 1048576.0, GR, R, U, SLACK, PF, t, gather / t / 1e9, rel_err(hout, href)); \
        fflush(stdout);                                                                                           \
    } while (0)
#define RUN4L(GR, R, U, SLACK, LW, LG)                                                                            \
    do {                                                                                                          \
        CK(hipMemset(d_out, 0, (size_t)N * ld * 4));                                                              \
        float t = time_ms([&] { launch_sweep4<GR, R, U>(a, B, d_done, SLACK, 0, 0, LW, LG); });                   \
        CK(hipMemcpy(hout.data(), d_out, hout.size() * 4, hipMemcpyDeviceToHost));                                \
        printf("K1s4 nb=%3u window %.2f MB group %d R=%d U=%d slack %d lag_w %d lag_g %d: %.3f ms  gather %.2f TB/s  err %.2e\n", \
               B.nb, (double)B.SB * GR * 16 / 1048576.0, GR, R, U, SLACK, LW, LG, t, gather / t / 1e9, rel_err(hout, href)); \
        fflush(stdout);                                                                                           \
    } while (0)
#define RUN5(GR, R, U, FORM)                                                                                      \
    do {                                                                                                          \
        CK(hipMemset(d_out, 0, (size_t)N * ld * 4));                                                              \
        CK(hipMemset(g_tacc, 0, 64));                                                                             \
        float t = time_ms([&] { launch_sweep5<GR, R, U, FORM>(a, B, d_done, 1, 0, 0); });                         \
        { unsigned long long h[5]; CK(hipMemcpy(h, g_tacc, 40, hipMemcpyDeviceToHost));                           \
          const double nwv = (double)h[4];                                                                        \
          printf("   per wave (mean over %.0f waves x 4 launches): gate %.2f ms, stage %.2f ms, rows+stage %.2f ms, post %.2f ms (100 MHz ticks)\n", nwv, \
                 h[0] / nwv * 1e-5, h[1] / nwv * 1e-5, h[2] / nwv * 1e-5, h[3] / nwv * 1e-5); }                    \
        CK(hipMemcpy(hout.data(), d_out, hout.size() * 4, hipMemcpyDeviceToHost));                                \
        printf("K1s5 nb=%3u window %.2f MB group %d R=%d U=%d form %d: %.3f ms  gather %.2f TB/s  err %.2e\n",    \
               B.nb, (double)B.SB * GR * 16 / 1048576.0, GR, R, U, FORM, t, gather / t / 1e9, rel_err(hout, href)); \
        fflush(stdout);                                                                                           \
    } while (0)
        RUN5(32, 10, 4, 0);
        if (false) {
            unsigned long long *d_dbg;
            const size_t nd = (size_t)2048 * 64 * 2;
            CK(hipMalloc(&d_dbg, nd * 8));
            CK(hipMemset(d_dbg, 0, nd * 8));
            const uint32_t Gs = 32u * sweep_occupancy<32, 8, 4>();
            launch_sweep<32, 8, 4>(a, B, Gs, d_done, 0, d_dbg);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> hd(nd);
            CK(hipMemcpy(hd.data(), d_dbg, nd * 8, hipMemcpyDeviceToHost));
            // sweep 0 of xcd 0: k in [0, Gs)
            const uint32_t nst = std::min(nb, 64u);
            unsigned long long t0 = ~0ull;
            for (uint32_t k = 0; k < Gs; ++k) t0 = std::min(t0, hd[((size_t)k * 64) * 2]);
            printf("dbg nb=%u Gs=%u (wall_clock64 ticks = 100 MHz -> 10 ns): per step: min/max start-of-wait, min/max end-of-wait, mean wait\n", nb, Gs);
            for (uint32_t b = 0; b < nst; b += (b < 6 ? 1 : 5)) {
                unsigned long long a0 = ~0ull, a1 = 0, e0 = ~0ull, e1 = 0; double w = 0;
                for (uint32_t k = 0; k < Gs; ++k) {
                    const unsigned long long ta = hd[((size_t)k * 64 + b) * 2] - t0, te = hd[((size_t)k * 64 + b) * 2 + 1] - t0;
                    a0 = std::min(a0, ta); a1 = std::max(a1, ta); e0 = std::min(e0, te); e1 = std::max(e1, te); w += (double)(te - ta);
                }
                printf("  step %2u: arrive %7.2f..%7.2f us  go %7.2f..%7.2f us  mean wait %6.2f us\n", b, a0 * 0.01, a1 * 0.01, e0 * 0.01, e1 * 0.01, w / Gs * 0.01);
            }
            {   // work time of workgroup k at step b = arrive[k][b+1] - go[k][b]
                std::vector<double> wk(Gs, 0.0);
                double mean_all = 0, mean_max = 0;
                for (uint32_t b = 2; b + 1 < nst; ++b) {
                    double mx = 0, mn = 0;
                    for (uint32_t k = 0; k < Gs; ++k) {
                        const double w = (double)(hd[((size_t)k * 64 + b + 1) * 2] - hd[((size_t)k * 64 + b) * 2 + 1]) * 0.01;
                        wk[k] += w; mx = std::max(mx, w); mn += w;
                    }
                    mean_all += mn / Gs; mean_max += mx;
                }
                const double ns = nst - 3;
                double kmin = 1e30, kmax = 0;
                for (uint32_t k = 0; k < Gs; ++k) { kmin = std::min(kmin, wk[k] / ns); kmax = std::max(kmax, wk[k] / ns); }
                printf("  work per step: mean over all %.2f us; mean of per-step max %.2f us; per-workgroup mean: min %.2f max %.2f us\n",
                       mean_all / ns, mean_max / ns, kmin, kmax);
                printf("  per-workgroup mean work by k (us):");
                for (uint32_t k = 0; k < Gs; ++k) printf("%s%.0f", k % 32 == 0 ? "\n    " : " ", wk[k] / ns);
                printf("\n");
            }
            CK(hipFree(d_dbg));
        }
        free_blocked(&B);
    }
    return 0;
}
