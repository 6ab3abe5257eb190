// gat_mh_blocked.hip -- multi-head GAT extension, source-blocked (L2-resident) forward and backward sweeps over
// K1b's blocked adjacency (spmm.hip).  Definition and row-wise kernels: gat_mh.hip; oracle: oracle/gat_mh_oracle.py.
#include "gat_mh.hpp"

namespace dory {

// partial rows are written once and read once by the reduce kernels: non-temporal, so that they do not push the source
// window the gathers live on out of the 4 MB L2
typedef float gm_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store4(float4 *p, const float4 &v) {
    __builtin_nontemporal_store((gm_v4f){v.x, v.y, v.z, v.w}, reinterpret_cast<gm_v4f *>(p));
}
__device__ __forceinline__ float4 nt_load4(const float4 *p) {
    const gm_v4f t = __builtin_nontemporal_load(reinterpret_cast<const gm_v4f *>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}

// ---- forward, source-blocked (K1b's idea applied to the attention-weighted sum) ---------------------------
// The row-wise kernel above gathers Z rows from all over the graph: every gather misses L2.  Here the softmax
// statistics come first (they only need el: N x K floats, L2-resident as a whole), then the weighted sum runs
// over K1b's source-blocked adjacency -- workgroup id -> XCD id & 7, each XCD walks one 5 MB window of Z rows at a
// time, alpha is recomputed per edge from el[src], er/m/den[dst] -- and a last kernel adds the partial rows and
// the self edge.  Same mapping as spmm_blocked_kernel (spmm.hip): GROUP lanes x float4 = one slab of a row.
template <bool GH>   // GH: ids >= N are ghost rows (partitioned run); without ghosts the select is compiled out
__global__ __launch_bounds__(256) void gatmh_stats_kernel(GatMhArgs a, const float *el, const float *elg,
                                                          const float *er, float *m_out, float *den_out) {
    const int lane = threadIdx.x & 63;
    const uint32_t v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= a.N) return;
    uint32_t KP = 1;
    while (KP < a.K) KP <<= 1;
    const uint32_t EPC = 64 / KP;
    const uint32_t k1 = lane % KP, j1 = lane / KP;
    const bool kok = k1 < a.K;
    const float er_v = kok ? er[(size_t)v * a.ldk + k1] : 0.f;
    const uint64_t e_beg = a.ptr[v], e_end = a.ptr[v + 1];   // edge e_end stands for the self edge
    float m = -INFINITY, den = 0.f;
    for (uint64_t e0 = e_beg; e0 <= e_end; e0 += EPC) {
        const uint64_t e = e0 + j1;
        if (e <= e_end && kok) {
            const uint32_t u = e < e_end ? a.idx[e] : v;
            const float el_u = (!GH || u < a.N) ? el[(size_t)u * a.ldk + k1] : elg[(size_t)(u - a.N) * a.ldk + k1];
            const float s = lrelu02(el_u + er_v);
            const float mn = fmaxf(m, s);
            den = den * __expf(m - mn) + __expf(s - mn);
            m = mn;
        }
    }
    for (uint32_t off = KP; off < 64; off <<= 1) {
        const float m2 = __shfl_xor(m, off, 64), d2 = __shfl_xor(den, off, 64);
        const float mn = fmaxf(m, m2);
        const float f1 = m == -INFINITY ? 0.f : __expf(m - mn), f2 = m2 == -INFINITY ? 0.f : __expf(m2 - mn);
        den = den * f1 + d2 * f2;
        m = mn;
    }
    if (j1 == 0 && kok) {
        m_out[(size_t)v * a.ldk + k1] = m;
        den_out[(size_t)v * a.ldk + k1] = den;
    }
}

#ifndef GATMH_BLK_ROWS_DEF
#define GATMH_BLK_ROWS_DEF 64
#endif
constexpr int GATMH_BLK_ROWS = GATMH_BLK_ROWS_DEF;   // destination rows per workgroup (as K1b)

// sum over the HL neighbouring lanes of a head (first two butterfly steps by DPP inside a quad: see the backward kernels)
// All inside the VALU: quad_perm [1,0,3,2] (0xB1) and [2,3,0,1] (0x4E) for the first two butterfly steps, then
// row_half_mirror (0x141: lane i <-> 7 - i of its 8 lanes) and row_mirror (0x140: lane i <-> 15 - i of its 16-lane DPP row)
// -- after the quad steps every lane of a quad holds the quad's total, so a mirror reaches "the other quad / the other half"
// as well as an xor would.  Only a head wider than 16 lanes (a single head on a 32-lane slab) still needs ds_bpermute.
// Lane groups of 16 are DPP rows (lane / 16), so this covers the single-head 64-float layer entirely (round 4; it used two
// ds_bpermute round trips through the LDS crossbar per entry before).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float head_lanes_sum(float v, int HL) {
    if (HL >= 2) v += dpp_mov<0xB1>(v);
    if (HL >= 4) v += dpp_mov<0x4E>(v);
    if (HL >= 8) v += dpp_mov<0x141>(v);
    if (HL >= 16) v += dpp_mov<0x140>(v);
    for (int o = 16; o < HL; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int GROUP, bool GH, bool ONLINE, bool ELFLY>
__global__ __launch_bounds__(256) void gatmh_forward_blocked_kernel(GatMhArgs a, BlockedAdj B, const float *z,
                                                                    const float *zg, const float *el,
                                                                    const float *elg, const float *er,
                                                                    const float *m_in, const float *den_in,
                                                                    float *partial, uint32_t tiles, uint32_t rounds,
                                                                    float *pm /*[nb][N][ldk]*/, float *pden,
                                                                    const float *a_l /*K x D, ELFLY*/) {
    constexpr int RPW = 64 / GROUP;
    constexpr int BLK_ITER = GATMH_BLK_ROWS / (4 * RPW);
    const uint32_t id = blockIdx.x;
    const uint32_t xcd = id & 7u;
    uint32_t q = id >> 3;
    const uint32_t tile_seq = q % tiles;
    q /= tiles;
    const uint32_t round = q % rounds;
    const uint32_t slab = q / rounds;
    const uint32_t b = round * 8u + xcd;
    if (b >= B.nb) return;
    const uint32_t tile = (uint32_t)(((uint64_t)tile_seq + (uint64_t)b * B.SB / GATMH_BLK_ROWS) % tiles);   // own tiles first (spmm.hip)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane % GROUP, gi = lane / GROUP;
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t col = slab * GROUP + li;
    const bool col_ok = col * 4 < a.K * a.D;
    const uint32_t ccol = col_ok ? col : 0;
    const uint32_t k = min((ccol * 4) / a.D, a.K - 1);          // one head per float4 (D % 4 == 0, or a single head)
    const float4 *z4 = reinterpret_cast<const float4 *>(z);
    const float4 *zg4 = reinterpret_cast<const float4 *>(zg);
    float4 *p4 = reinterpret_cast<float4 *>(partial) + (size_t)b * a.N * nchunk;
    const uint32_t *boff = B.boff + (size_t)b * (a.N + 1);
    const uint64_t base = B.bbase[b];
    float4 al4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int HL = a.K == 1 ? GROUP : (int)(a.D >> 2);    // lanes per head (ELFLY: 1, 2, 4, or the 16 of a single-head row)
    if constexpr (ELFLY) {   // a_l is a dense K x D vector: the last float4 of a 41-feature head is read element by element
        const uint32_t f0 = ccol * 4, KD = a.K * a.D;
        if (col_ok) al4 = make_float4(a_l[f0], f0 + 1 < KD ? a_l[f0 + 1] : 0.f, f0 + 2 < KD ? a_l[f0 + 2] : 0.f, f0 + 3 < KD ? a_l[f0 + 3] : 0.f);
    }
    auto el_of = [&](const float4 &x) -> float {          // ELFLY: this lane's head's score of the gathered row
        return head_lanes_sum(fmaf(x.x, al4.x, fmaf(x.y, al4.y, fmaf(x.z, al4.z, x.w * al4.w))), HL);
    };
#pragma unroll 1
    for (int it = 0; it < BLK_ITER; ++it) {
        const uint32_t v = tile * GATMH_BLK_ROWS + (uint32_t)((it * 4 + wave) * RPW + gi);
        const bool row_ok = v < a.N;
        const uint32_t vv = row_ok ? v : 0;
        uint64_t e = row_ok ? base + boff[v] : 0;
        const uint64_t end = row_ok ? base + boff[v + 1] : 0;
        const float er_v = er[(size_t)vv * a.ldk + k];
        float m_v = -INFINITY, idn = 1.f, den = 0.f;
        if constexpr (!ONLINE) {
            m_v = m_in[(size_t)vv * a.ldk + k];
            idn = 1.f / den_in[(size_t)vv * a.ldk + k];
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // one entry: ONLINE moves the running maximum first (acc and den follow it), then adds exp(s - m) * x
        auto add_entry = [&](float el_s, const float4 &x) {
            const float sc = lrelu02(el_s + er_v);
            float al;
            if constexpr (ONLINE) {
                const float mn = fmaxf(m_v, sc);
                const float f = __expf(m_v - mn);          // (exp(-inf) = 0 on the first entry: acc and den are 0 anyway)
                al = __expf(sc - mn);
                acc.x *= f; acc.y *= f; acc.z *= f; acc.w *= f;
                den = fmaf(den, f, al);
                m_v = mn;
            } else {
                al = __expf(sc - m_v) * idn;
            }
            acc.x = fmaf(al, x.x, acc.x); acc.y = fmaf(al, x.y, acc.y);
            acc.z = fmaf(al, x.z, acc.z); acc.w = fmaf(al, x.w, acc.w);
        };
        while (e < end) {
            const int n = (end - e) < (uint64_t)GROUP ? (int)(end - e) : GROUP;
            uint32_t my_idx = 0;
            if (li < n) my_idx = __builtin_nontemporal_load(B.bidx + e + li);
            int j = 0;
            for (; j + 4 <= n; j += 4) {
                float4 x[4];
                float w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t s = (uint32_t)__shfl((int)my_idx, j + u, GROUP);
                    x[u] = ((!GH || s < a.N) ? z4 + (size_t)s * nchunk : zg4 + (size_t)(s - a.N) * nchunk)[ccol];
                    if constexpr (!ELFLY) w[u] = (!GH || s < a.N) ? el[(size_t)s * a.ldk + k] : elg[(size_t)(s - a.N) * a.ldk + k];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) add_entry(ELFLY ? el_of(x[u]) : w[u], x[u]);
            }
            for (; j < n; ++j) {
                const uint32_t s = (uint32_t)__shfl((int)my_idx, j, GROUP);
                const float4 x = ((!GH || s < a.N) ? z4 + (size_t)s * nchunk : zg4 + (size_t)(s - a.N) * nchunk)[ccol];
                float el_s;
                if constexpr (ELFLY) el_s = el_of(x);
                else el_s = (!GH || s < a.N) ? el[(size_t)s * a.ldk + k] : elg[(size_t)(s - a.N) * a.ldk + k];
                add_entry(el_s, x);
            }
            e += n;
        }
        if (row_ok && col_ok) {
            nt_store4(p4 + (size_t)v * nchunk + col, acc);
            if constexpr (ONLINE) {   // the first float4 of a head carries its statistics (every lane of a head holds the same pair)
                if ((ccol * 4) % a.D < 4 || a.K == 1) {
                    const size_t o_ = ((size_t)b * a.N + v) * a.ldk + k;
                    pm[o_] = m_v;
                    pden[o_] = den;
                }
            }
        }
    }
}

// o[v,:] = sum_b partial[b][v,:] + alpha_self * z[v,:]
// ONLINE: the blocks' partial rows are unnormalised sums against their own maxima m_b: o = (sum_b exp(m_b - m) partial_b +
// exp(s_self - m) z_v) / (sum_b exp(m_b - m) den_b + exp(s_self - m)), m = max(max_b m_b, s_self); m and den are written
// here (the backward sweeps read them).  Block order, like the plain sum.
template <bool ONLINE>
__global__ __launch_bounds__(256) void gatmh_forward_reduce_kernel(GatMhArgs a, uint32_t nb, const float *partial,
                                                                   const float *z, const float *el, const float *er,
                                                                   float *m_io, float *den_io, float *o,
                                                                   const float *pm, const float *pden) {
    const uint32_t nchunk = a.ld >> 2;
    const size_t n = (size_t)a.N * nchunk;
    const float4 *p4 = reinterpret_cast<const float4 *>(partial);
    const float4 *z4 = reinterpret_cast<const float4 *>(z);
    float4 *o4 = reinterpret_cast<float4 *>(o);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t v = (uint32_t)(i / nchunk), col = (uint32_t)(i % nchunk);
        if (col * 4 >= a.K * a.D) continue;
        const uint32_t k = min((col * 4) / a.D, a.K - 1);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const size_t vk = (size_t)v * a.ldk + k;
        const float4 x = z4[i];
        if constexpr (ONLINE) {
            const float s_self = lrelu02(el[vk] + er[vk]);
            const size_t nk = (size_t)a.N * a.ldk;
            float m = s_self;
            for (uint32_t b = 0; b < nb; ++b) m = fmaxf(m, pm[(size_t)b * nk + vk]);
            float den = 0.f;
            for (uint32_t b = 0; b < nb; ++b) {
                const float f = __expf(pm[(size_t)b * nk + vk] - m);          // an empty segment left m_b = -inf: weight 0
                const float4 p = nt_load4(p4 + (size_t)b * n + i);
                acc.x = fmaf(f, p.x, acc.x); acc.y = fmaf(f, p.y, acc.y); acc.z = fmaf(f, p.z, acc.z); acc.w = fmaf(f, p.w, acc.w);
                den = fmaf(f, pden[(size_t)b * nk + vk], den);
            }
            const float fs = __expf(s_self - m);
            den += fs;
            acc.x = fmaf(fs, x.x, acc.x); acc.y = fmaf(fs, x.y, acc.y); acc.z = fmaf(fs, x.z, acc.z); acc.w = fmaf(fs, x.w, acc.w);
            const float idn = 1.f / den;
            acc.x *= idn; acc.y *= idn; acc.z *= idn; acc.w *= idn;
            if ((col * 4) % a.D < 4 || a.K == 1) { m_io[vk] = m; den_io[vk] = den; }
        } else {
            for (uint32_t b = 0; b < nb; ++b) {
                const float4 p = nt_load4(p4 + (size_t)b * n + i);
                acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
            }
            const float al = __expf(lrelu02(el[vk] + er[vk]) - m_io[vk]) / den_io[vk];   // the self edge
            acc.x = fmaf(al, x.x, acc.x); acc.y = fmaf(al, x.y, acc.y);
            acc.z = fmaf(al, x.z, acc.z); acc.w = fmaf(al, x.w, acc.w);
        }
        o4[i] = acc;
    }
}

hipError_t launch_gatmh_forward_blocked(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk,
                                        const uint64_t *colptr, const uint32_t *rowidx, const BlockedAdj &B,
                                        const float *z, const float *zg, const float *el, const float *elg,
                                        const float *er, float *o, float *m, float *den, float *partial, bool ghosts,
                                        hipStream_t s, float *stat_partial /* 2 x nb x N x ldk floats: the blocks' (m_b, den_b);
                                        nullptr = separate statistics pass first (round 1-3 form) */,
                                        const float *a_l /* K x D attention vector: scores of the sources formed from the gathered
                                        rows (heads of <= 16 features, 32-lane launches); nullptr = gather el */) {
    if (N == 0) return hipSuccess;
    if (!gatmh_shape_ok(K, D) || ((D & 3) && K != 1) || (ld & 3) || B.nb == 0) return hipErrorInvalidValue;
    GatMhArgs a{N, K, D, ld, ldk, colptr, rowidx};
    const bool online = stat_partial != nullptr;
    float *pm = stat_partial, *pden = online ? stat_partial + (size_t)B.nb * N * ldk : nullptr;
    if (!online) {
        if (ghosts) hipLaunchKernelGGL(gatmh_stats_kernel<true>, dim3((N + 3) / 4), dim3(256), 0, s, a, el, elg, er, m, den);
        else hipLaunchKernelGGL(gatmh_stats_kernel<false>, dim3((N + 3) / 4), dim3(256), 0, s, a, el, elg, er, m, den);
    }
    const uint32_t nchunk = ld >> 2;
    const int group = ld >= 128 ? 32 : 16;
    const uint32_t slabs = (((K * D + 3) >> 2) + group - 1) / group;
    const uint32_t tiles = (N + GATMH_BLK_ROWS - 1) / GATMH_BLK_ROWS, rounds = (B.nb + 7) / 8;
    const uint64_t grid = (uint64_t)slabs * rounds * tiles * 8;
    if (grid > 0x7FFFFFFFull) return hipErrorInvalidValue;
#define GATMH_FWD(G, H)                                                                                                  \
    do {                                                                                                                 \
        if (online) hipLaunchKernelGGL((gatmh_forward_blocked_kernel<G, H, true, false>), dim3((uint32_t)grid), dim3(256), 0, s, a, B, z, zg, \
                                       el, elg, er, m, den, partial, tiles, rounds, pm, pden, a_l);                      \
        else hipLaunchKernelGGL((gatmh_forward_blocked_kernel<G, H, false, false>), dim3((uint32_t)grid), dim3(256), 0, s, a, B, z, zg, el,   \
                                elg, er, m, den, partial, tiles, rounds, pm, pden, a_l);                                 \
    } while (0)
    // el from the gathered row: heads of one quad of lanes (D <= 16) on 32-lane slabs, or the single head of a one-slab 16-lane row
    const bool elfly32 = a_l != nullptr && online && group == 32 && K > 1 && (D == 4 || D == 8 || D == 16);
    const bool elfly16 = a_l != nullptr && online && group == 16 && K == 1 && slabs == 1;
#define GATMH_FWD_EL(G, H)                                                                                               \
    hipLaunchKernelGGL((gatmh_forward_blocked_kernel<G, H, true, true>), dim3((uint32_t)grid), dim3(256), 0, s, a, B, z, zg, el, elg, \
                       er, m, den, partial, tiles, rounds, pm, pden, a_l)
    if (elfly32 && ghosts) GATMH_FWD_EL(32, true);
    else if (elfly32) GATMH_FWD_EL(32, false);
    else if (elfly16 && ghosts) GATMH_FWD_EL(16, true);
    else if (elfly16) GATMH_FWD_EL(16, false);
    else if (group == 32 && ghosts) GATMH_FWD(32, true);
    else if (group == 32) GATMH_FWD(32, false);
    else if (ghosts) GATMH_FWD(16, true);
    else GATMH_FWD(16, false);
#undef GATMH_FWD
#undef GATMH_FWD_EL
    const size_t n = (size_t)N * nchunk;
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    if (online) hipLaunchKernelGGL(gatmh_forward_reduce_kernel<true>, dim3(blocks), dim3(256), 0, s, a, B.nb, partial, z, el, er, m, den, o, pm, pden);
    else hipLaunchKernelGGL(gatmh_forward_reduce_kernel<false>, dim3(blocks), dim3(256), 0, s, a, B.nb, partial, z, el, er, m, den, o, pm, pden);
    return hipGetLastError();
}

// ---- backward, source-blocked (same idea as gatmh_forward_blocked_kernel) ------------------------------------
// K1b's lane mapping: GROUP lanes x float4 cover one 128- or 64-float slab of a row, HL = D/4 neighbouring lanes share
// a head (a single head spans the whole group), so <dO[v,k,:], Z[u,k,:]> is a 4-term dot per lane plus log2(HL)
// xor-shuffles inside the head.  Destination side: per-(block, v, k) partial (t, a1, a2); source side: per-block
// partial dZ rows and partial del; small reduce kernels add the blocks in order, then the self edge.
// sum over the HL neighbouring lanes of a head.  The first two butterfly steps stay inside a quad: DPP quad_perm
// ([1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E) moves the operand in the VALU instead of a ds_bpermute round trip through LDS
template <int GROUP, bool GH, bool ELFLY /* el[src] from the gathered row (forward kernel above) */>
__global__ __launch_bounds__(256) void gatmh_bwd_dst_blocked_kernel(GatMhArgs a, BlockedAdj B, const float *z,
                                                                    const float *zg, const float *el,
                                                                    const float *elg, const float *er,
                                                                    const float *m_in, const float *den_in,
                                                                    const float *d_o, float4 *pst /*[nb][N][K]*/,
                                                                    uint32_t tiles, uint32_t rounds, int HL, const float *a_l) {
    constexpr int RPW = 64 / GROUP;
    constexpr int BLK_ITER = GATMH_BLK_ROWS / (4 * RPW);
    const uint32_t id = blockIdx.x;
    const uint32_t xcd = id & 7u;
    uint32_t q = id >> 3;
    const uint32_t tile_seq = q % tiles;
    q /= tiles;
    const uint32_t round = q % rounds;
    const uint32_t slab = q / rounds;
    const uint32_t b = round * 8u + xcd;
    if (b >= B.nb) return;
    const uint32_t tile = (uint32_t)(((uint64_t)tile_seq + (uint64_t)b * B.SB / GATMH_BLK_ROWS) % tiles);   // own tiles first (spmm.hip)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane % GROUP, gi = lane / GROUP;
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t col = slab * GROUP + li;
    const bool col_ok = col * 4 < a.K * a.D;
    const uint32_t ccol = col_ok ? col : 0;
    const uint32_t k = min((ccol * 4) / a.D, a.K - 1);
    const float4 *z4 = reinterpret_cast<const float4 *>(z);
    const float4 *zg4 = reinterpret_cast<const float4 *>(zg);
    const float4 *do4 = reinterpret_cast<const float4 *>(d_o);
    const uint32_t *boff = B.boff + (size_t)b * (a.N + 1);
    const uint64_t base = B.bbase[b];
    float4 al4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (ELFLY) {
        const uint32_t f0 = ccol * 4, KD = a.K * a.D;
        if (col_ok) al4 = make_float4(a_l[f0], f0 + 1 < KD ? a_l[f0 + 1] : 0.f, f0 + 2 < KD ? a_l[f0 + 2] : 0.f, f0 + 3 < KD ? a_l[f0 + 3] : 0.f);
    }
#pragma unroll 1
    for (int it = 0; it < BLK_ITER; ++it) {
        const uint32_t v = tile * GATMH_BLK_ROWS + (uint32_t)((it * 4 + wave) * RPW + gi);
        const bool row_ok = v < a.N;
        const uint32_t vv = row_ok ? v : 0;
        uint64_t e = row_ok ? base + boff[v] : 0;
        const uint64_t end = row_ok ? base + boff[v + 1] : 0;
        float4 dv = do4[(size_t)vv * nchunk + ccol];
        if (!col_ok) dv = make_float4(0.f, 0.f, 0.f, 0.f);
        const float er_v = er[(size_t)vv * a.ldk + k], m_v = m_in[(size_t)vv * a.ldk + k];
        const float idn = 1.f / den_in[(size_t)vv * a.ldk + k];
        float t = 0.f, a1 = 0.f, a2 = 0.f;
        while (e < end) {
            const int n = (end - e) < (uint64_t)GROUP ? (int)(end - e) : GROUP;
            uint32_t my_idx = 0;
            if (li < n) my_idx = __builtin_nontemporal_load(B.bidx + e + li);
            for (int j = 0; j < n; j += 4) {
                float4 x[4];
                float w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jj = j + u < n ? j + u : n - 1;      // dead slots repeat the last edge, weight 0 below
                    const uint32_t s = (uint32_t)__shfl((int)my_idx, jj, GROUP);
                    x[u] = ((!GH || s < a.N) ? z4 + (size_t)s * nchunk : zg4 + (size_t)(s - a.N) * nchunk)[ccol];
                    if constexpr (!ELFLY) w[u] = (!GH || s < a.N) ? el[(size_t)s * a.ldk + k] : elg[(size_t)(s - a.N) * a.ldk + k];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float da = dv.x * x[u].x;
                    da = fmaf(dv.y, x[u].y, da); da = fmaf(dv.z, x[u].z, da); da = fmaf(dv.w, x[u].w, da);
                    da = head_lanes_sum(da, HL);
                    if constexpr (ELFLY)
                        w[u] = head_lanes_sum(fmaf(x[u].x, al4.x, fmaf(x[u].y, al4.y, fmaf(x[u].z, al4.z, x[u].w * al4.w))), HL);
                    const float pre = w[u] + er_v;
                    const float al = j + u < n ? __expf(lrelu02(pre) - m_v) * idn : 0.f;
                    const float lp = pre > 0.f ? 1.f : GATMH_SLOPE;
                    t = fmaf(al, da, t);
                    a1 = fmaf(al * da, lp, a1);
                    a2 = fmaf(al, lp, a2);
                }
            }
            e += n;
        }
        if (row_ok && col_ok && (li % HL) == 0) nt_store4(pst + ((size_t)b * a.N + v) * a.K + k, make_float4(t, a1, a2, 0.f));
    }
}

// t, der, st4 = (er, m, 1/den, t): blocks in order, then the self edge
__global__ void gatmh_bwd_dst_reduce_kernel(GatMhArgs a, uint32_t nb, const float4 *pst, const float *z, const float *el,
                                            const float *er, const float *m_in, const float *den_in, const float *d_o,
                                            float *t_out, float *der_out, float4 *st4, uint32_t lds4) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)a.N * a.K) return;
    const uint32_t v = (uint32_t)(i / a.K), k = (uint32_t)(i % a.K);
    float t = 0.f, a1 = 0.f, a2 = 0.f;
    for (uint32_t b = 0; b < nb; ++b) {
        const float4 p = nt_load4(pst + ((size_t)b * a.N + v) * a.K + k);
        t += p.x; a1 += p.y; a2 += p.z;
    }
    const float *zr = z + (size_t)v * a.ld + (size_t)k * a.D, *dr = d_o + (size_t)v * a.ld + (size_t)k * a.D;
    float da = 0.f;
    for (uint32_t d = 0; d < a.D; ++d) da = fmaf(dr[d], zr[d], da);
    const size_t vk = (size_t)v * a.ldk + k;
    const float idn = 1.f / den_in[vk];
    const float pre = el[vk] + er[vk];
    const float al = __expf(lrelu02(pre) - m_in[vk]) * idn;
    const float lp = pre > 0.f ? 1.f : GATMH_SLOPE;
    t = fmaf(al, da, t);
    a1 = fmaf(al * da, lp, a1);
    a2 = fmaf(al, lp, a2);
    t_out[vk] = t;
    der_out[vk] = a1 - t * a2;
    st4[(size_t)v * lds4 + k] = make_float4(er[vk], m_in[vk], idn, t);
}

template <int GROUP, bool GH>
__global__ __launch_bounds__(256) void gatmh_bwd_src_blocked_kernel(GatMhArgs a, BlockedAdj B, const float *z,
                                                                    const float *el, const float4 *st4,
                                                                    const float4 *stg, uint32_t lds4, const float *d_o,
                                                                    const float *dog, float *pdz /*[nb][N][ld]*/,
                                                                    float *pdel /*[nb][N][K]*/, uint32_t tiles,
                                                                    uint32_t rounds, int HL) {
    constexpr int RPW = 64 / GROUP;
    constexpr int BLK_ITER = GATMH_BLK_ROWS / (4 * RPW);
    const uint32_t id = blockIdx.x;
    const uint32_t xcd = id & 7u;
    uint32_t q = id >> 3;
    const uint32_t tile_seq = q % tiles;
    q /= tiles;
    const uint32_t round = q % rounds;
    const uint32_t slab = q / rounds;
    const uint32_t b = round * 8u + xcd;
    if (b >= B.nb) return;
    const uint32_t tile = (uint32_t)(((uint64_t)tile_seq + (uint64_t)b * B.SB / GATMH_BLK_ROWS) % tiles);   // own tiles first (spmm.hip)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane % GROUP, gi = lane / GROUP;
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t col = slab * GROUP + li;
    const bool col_ok = col * 4 < a.K * a.D;
    const uint32_t ccol = col_ok ? col : 0;
    const uint32_t k = min((ccol * 4) / a.D, a.K - 1);
    const float4 *z4 = reinterpret_cast<const float4 *>(z);
    const float4 *do4 = reinterpret_cast<const float4 *>(d_o);
    const float4 *dog4 = reinterpret_cast<const float4 *>(dog);
    float4 *p4 = reinterpret_cast<float4 *>(pdz) + (size_t)b * a.N * nchunk;
    const uint32_t *boff = B.boff + (size_t)b * (a.N + 1);
    const uint64_t base = B.bbase[b];
#pragma unroll 1
    for (int it = 0; it < BLK_ITER; ++it) {
        const uint32_t u = tile * GATMH_BLK_ROWS + (uint32_t)((it * 4 + wave) * RPW + gi);
        const bool row_ok = u < a.N;
        const uint32_t uu = row_ok ? u : 0;
        uint64_t e = row_ok ? base + boff[u] : 0;
        const uint64_t end = row_ok ? base + boff[u + 1] : 0;
        float4 zu = z4[(size_t)uu * nchunk + ccol];
        if (!col_ok) zu = make_float4(0.f, 0.f, 0.f, 0.f);
        const float el_u = el[(size_t)uu * a.ldk + k];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float del = 0.f;
        while (e < end) {
            const int n = (end - e) < (uint64_t)GROUP ? (int)(end - e) : GROUP;
            uint32_t my_idx = 0;
            if (li < n) my_idx = __builtin_nontemporal_load(B.bidx + e + li);
            for (int j = 0; j < n; j += 4) {
                float4 x[4], sv[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int jj = j + c < n ? j + c : n - 1;
                    const uint32_t v = (uint32_t)__shfl((int)my_idx, jj, GROUP);   // ids >= N: ghost destinations
                    x[c] = ((!GH || v < a.N) ? do4 + (size_t)v * nchunk : dog4 + (size_t)(v - a.N) * nchunk)[ccol];
                    sv[c] = (!GH || v < a.N) ? st4[(size_t)v * lds4 + k] : stg[(size_t)(v - a.N) * lds4 + k];
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float da = zu.x * x[c].x;
                    da = fmaf(zu.y, x[c].y, da); da = fmaf(zu.z, x[c].z, da); da = fmaf(zu.w, x[c].w, da);
                    da = head_lanes_sum(da, HL);
                    const float pre = el_u + sv[c].x;
                    const float al = j + c < n ? __expf(lrelu02(pre) - sv[c].y) * sv[c].z : 0.f;
                    const float lp = pre > 0.f ? 1.f : GATMH_SLOPE;
                    del = fmaf(al * (da - sv[c].w), lp, del);
                    acc.x = fmaf(al, x[c].x, acc.x); acc.y = fmaf(al, x[c].y, acc.y);
                    acc.z = fmaf(al, x[c].z, acc.z); acc.w = fmaf(al, x[c].w, acc.w);
                }
            }
            e += n;
        }
        if (row_ok && col_ok) {
            nt_store4(p4 + (size_t)u * nchunk + col, acc);
            if ((li % HL) == 0) __builtin_nontemporal_store(del, pdel + ((size_t)b * a.N + u) * a.K + k);
        }
    }
}

// del[u,k] = sum_b pdel + self edge
__global__ void gatmh_bwd_del_reduce_kernel(GatMhArgs a, uint32_t nb, const float *pdel, const float *z, const float *el,
                                            const float4 *st4, uint32_t lds4, const float *d_o, float *del_out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)a.N * a.K) return;
    const uint32_t u = (uint32_t)(i / a.K), k = (uint32_t)(i % a.K);
    float del = 0.f;
    for (uint32_t b = 0; b < nb; ++b) del += __builtin_nontemporal_load(pdel + ((size_t)b * a.N + u) * a.K + k);
    const float *zr = z + (size_t)u * a.ld + (size_t)k * a.D, *dr = d_o + (size_t)u * a.ld + (size_t)k * a.D;
    float da = 0.f;
    for (uint32_t d = 0; d < a.D; ++d) da = fmaf(dr[d], zr[d], da);
    const float4 sv = st4[(size_t)u * lds4 + k];
    const float pre = el[(size_t)u * a.ldk + k] + sv.x;
    const float al = __expf(lrelu02(pre) - sv.y) * sv.z;
    const float lp = pre > 0.f ? 1.f : GATMH_SLOPE;
    del = fmaf(al * (da - sv.w), lp, del);
    del_out[(size_t)u * a.ldk + k] = del;
}

// dz[u,:] = sum_b pdz[b][u,:] + alpha_self dO[u,:] + del[u,k] a_l + der[u,k] a_r
__global__ __launch_bounds__(256) void gatmh_bwd_dz_reduce_kernel(GatMhArgs a, uint32_t nb, const float *pdz,
                                                                  const float *el, const float4 *st4, uint32_t lds4,
                                                                  const float *d_o, const float *del, const float *der,
                                                                  const float *a_l,
                                                                  const float *a_r, float *dz) {
    const uint32_t nchunk = a.ld >> 2;
    const size_t n = (size_t)a.N * nchunk;
    const float4 *p4 = reinterpret_cast<const float4 *>(pdz);
    const float4 *do4 = reinterpret_cast<const float4 *>(d_o);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t u = (uint32_t)(i / nchunk), col = (uint32_t)(i % nchunk);
        if (col * 4 >= a.K * a.D) continue;
        const uint32_t k = min((col * 4) / a.D, a.K - 1);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t b = 0; b < nb; ++b) {
            const float4 p = nt_load4(p4 + (size_t)b * n + i);
            acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        }
        const float4 sv = st4[(size_t)u * lds4 + k];
        const float al = __expf(lrelu02(el[(size_t)u * a.ldk + k] + sv.x) - sv.y) * sv.z;   // self edge
        const float4 x = do4[i];
        const float dl = del[(size_t)u * a.ldk + k], dr = der[(size_t)u * a.ldk + k];
        float r[4] = {fmaf(al, x.x, acc.x), fmaf(al, x.y, acc.y), fmaf(al, x.z, acc.z), fmaf(al, x.w, acc.w)};
        float *out = dz + (size_t)u * a.ld + (size_t)col * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t f = col * 4 + c;
            if (f < a.K * a.D) out[c] = r[c] + dl * a_l[f] + dr * a_r[f];
        }
    }
}

// conditions of the blocked backward: one slab per head group, heads aligned to float4 lanes
static int gatmh_blocked_hl(uint32_t K, uint32_t D, uint32_t ld) {
    const int group = ld >= 128 ? 32 : 16;
    if (K == 1) return (ld <= (uint32_t)group * 4) ? group : 0;              // a single head: the whole (one-slab) row
    if ((D & 3) || (D & (D - 1)) || D / 4 > (uint32_t)group) return 0;
    return (int)(D / 4);
}

// The backward sweep in its two phases.  Between them a partitioned run exchanges the ghost rows of dO and st4
// (destinations of local out-edges owned elsewhere); a single partition just calls both.
struct GatMhBwdPlan {
    int HL, group;
    uint32_t slabs, tiles;
    int nk_blocks, row_blocks;
};
static bool gatmh_bwd_plan(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, GatMhBwdPlan *p) {
    p->HL = gatmh_blocked_hl(K, D, ld);
    if (!p->HL || !gatmh_shape_ok(K, D) || 4 * K > ld) return false;
    p->group = ld >= 128 ? 32 : 16;
    p->slabs = (((K * D + 3) >> 2) + p->group - 1) / p->group;
    p->tiles = (N + GATMH_BLK_ROWS - 1) / GATMH_BLK_ROWS;
    p->nk_blocks = (int)(((uint64_t)N * K + 255) / 256);
    const size_t nrow4 = (size_t)N * (ld >> 2);
    p->row_blocks = (int)((nrow4 + 255) / 256 < 8192 ? (nrow4 + 255) / 256 : 8192);
    return true;
}

// phase A, destination side over the in-edges: t, der and st4 = (er, m, 1/den, t) of every local (v, k)
hipError_t launch_gatmh_backward_blocked_dst(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk,
                                             const BlockedAdj &Bin, const float *z, const float *zg, const float *el,
                                             const float *elg, const float *er, const float *m, const float *den,
                                             const float *d_o, float *t, float *der, float *partial, float4 *st4,
                                             uint32_t lds4, bool ghosts, hipStream_t s, const float *a_l) {
    if (N == 0) return hipSuccess;
    GatMhBwdPlan p;
    if (!gatmh_bwd_plan(N, K, D, ld, &p) || Bin.nb == 0) return hipErrorInvalidValue;
    GatMhArgs a{N, K, D, ld, ldk, nullptr, nullptr};
    const uint32_t rounds = (Bin.nb + 7) / 8;
    const uint64_t grid = (uint64_t)p.slabs * rounds * p.tiles * 8;
    if (grid > 0x7FFFFFFFull) return hipErrorInvalidValue;
    float4 *pst = reinterpret_cast<float4 *>(partial);
#define GATMH_DST(G, H)                                                                                                  \
    hipLaunchKernelGGL((gatmh_bwd_dst_blocked_kernel<G, H, false>), dim3((uint32_t)grid), dim3(256), 0, s, a, Bin, z, zg, el, elg,  \
                       er, m, den, d_o, pst, p.tiles, rounds, p.HL, a_l)
    const bool elfly32 = a_l != nullptr && p.group == 32 && K > 1 && (D == 4 || D == 8 || D == 16);
    const bool elfly16 = a_l != nullptr && p.group == 16 && K == 1 && p.slabs == 1;
#define GATMH_DST_EL(G, H)                                                                                               \
    hipLaunchKernelGGL((gatmh_bwd_dst_blocked_kernel<G, H, true>), dim3((uint32_t)grid), dim3(256), 0, s, a, Bin, z, zg, el, elg,   \
                       er, m, den, d_o, pst, p.tiles, rounds, p.HL, a_l)
    if (elfly32 && ghosts) GATMH_DST_EL(32, true);
    else if (elfly32) GATMH_DST_EL(32, false);
    else if (elfly16 && ghosts) GATMH_DST_EL(16, true);
    else if (elfly16) GATMH_DST_EL(16, false);
    else if (p.group == 32 && ghosts) GATMH_DST(32, true);
    else if (p.group == 32) GATMH_DST(32, false);
    else if (ghosts) GATMH_DST(16, true);
    else GATMH_DST(16, false);
#undef GATMH_DST
#undef GATMH_DST_EL
    hipLaunchKernelGGL(gatmh_bwd_dst_reduce_kernel, dim3(p.nk_blocks), dim3(256), 0, s, a, Bin.nb, pst, z, el, er, m, den, d_o,
                       t, der, st4, lds4);
    return hipGetLastError();
}

// phase B, source side over the out-edges: del and dz (needs dO and st4 of the ghost destinations too)
hipError_t launch_gatmh_backward_blocked_src(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk,
                                             const BlockedAdj &Bout, const float *z, const float *el, const float *d_o,
                                             const float *dog, const float4 *st4, const float4 *stg, uint32_t lds4,
                                             const float *der, const float *a_l, const float *a_r, float *del, float *dz,
                                             float *partial, bool ghosts, hipStream_t s) {
    if (N == 0) return hipSuccess;
    GatMhBwdPlan p;
    if (!gatmh_bwd_plan(N, K, D, ld, &p) || Bout.nb == 0) return hipErrorInvalidValue;
    GatMhArgs a{N, K, D, ld, ldk, nullptr, nullptr};
    const uint32_t rounds = (Bout.nb + 7) / 8;
    const uint64_t grid = (uint64_t)p.slabs * rounds * p.tiles * 8;
    if (grid > 0x7FFFFFFFull) return hipErrorInvalidValue;
    float *pdz = partial, *pdel = partial + (size_t)Bout.nb * N * ld;
#define GATMH_SRC(G, H)                                                                                                  \
    hipLaunchKernelGGL((gatmh_bwd_src_blocked_kernel<G, H>), dim3((uint32_t)grid), dim3(256), 0, s, a, Bout, z, el, st4, stg,  \
                       lds4, d_o, dog, pdz, pdel, p.tiles, rounds, p.HL)
    if (p.group == 32 && ghosts) GATMH_SRC(32, true);
    else if (p.group == 32) GATMH_SRC(32, false);
    else if (ghosts) GATMH_SRC(16, true);
    else GATMH_SRC(16, false);
#undef GATMH_SRC
    hipLaunchKernelGGL(gatmh_bwd_del_reduce_kernel, dim3(p.nk_blocks), dim3(256), 0, s, a, Bout.nb, pdel, z, el, st4, lds4, d_o,
                       del);
    hipLaunchKernelGGL(gatmh_bwd_dz_reduce_kernel, dim3(p.row_blocks), dim3(256), 0, s, a, Bout.nb, pdz, el, st4, lds4, d_o, del,
                       der, a_l, a_r, dz);
    return hipGetLastError();
}

bool gatmh_backward_blocked_ok(uint32_t K, uint32_t D, uint32_t ld) { return gatmh_blocked_hl(K, D, ld) != 0 && 4 * K <= ld; }

}  // namespace dory
