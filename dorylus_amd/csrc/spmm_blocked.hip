// spmm_blocked.hip -- K1b: the source-blocked SpMM with one partial row per (source block, output row) and a reduce
// kernel.  Round 1's L2-resident form; since round 2 the default is K1s (spmm.hip).  Kept for two users: the multi-head
// GAT kernels (gat_mh_blocked.hip), which walk K1b's blocked adjacency (build_blocked) with their own per-edge
// arithmetic, and option spmm_variant = 1.  Same reference as spmm.hip: Engine::aggregateGCN / aggregateGAT
// (src/graph-server/engine/ops/gcn_ops.cpp:130-191, gat_ops.cpp:173-243).
#include <algorithm>
#include <cmath>
#include <vector>

#include "spmm_common.hpp"

namespace dory {

// =======================================================================================
// K1b: source-blocked, XCD-aware variant -- gathers served from the 4 MB per-XCD L2.
//
// Measured on MI355X (profiles/r01_spmm_l2_window_probe.txt): the row gather of K1 runs
// at ~7.4 TB/s whenever the source operand does not fit L2 (every miss crosses the
// fabric; rocprof FETCH_SIZE = E*ld*4, no reuse), but at 18-20 TB/s when the sources of
// all concurrently running workgroups of an XCD lie inside a <= 4 MB window.
// On a random graph that window has to be imposed:
//   * the virtual source space [local rows ; ghost rows] is cut into nb = 8k blocks of SB
//     rows; the edges are regrouped once (build_blocked) into a per-block CSR
//     (block-major, row-minor, original edge order inside a (block,row) segment);
//   * features are processed in slabs of W = 64 floats (256 B per row), so one block's
//     working set is SB * 256 B (3.7 MB at Reddit scale, nb = 16);
//   * workgroup id -> (xcd = id % 8, slab, round, tile): block b = round*8 + xcd, so the
//     workgroups the hardware places on XCD x (observed: id % 8; used for speed only,
//     never for correctness) all gather from block b's window at the same time;
//   * each (block,row) segment yields a partial row slab, written once to
//     partial[b][v][slab]; spmm_reduce_kernel adds self term + partials in block order.
// The price is one partial buffer of nb * N * ld floats written and read once, and the
// index arrays re-read once per slab -- streamed HBM traffic that replaces the E*ld*4 B
// of random fabric traffic (profiles/ has the FETCH_SIZE before/after).
// Summation order: by block, then edge order inside the block (deterministic; differs
// from the pure edge order of K1 / the CPU loop by fp32 reassociation, ~1e-7 relative).
// Applies when the values are static (GCN) and nb stays small; callers fall back to K1.
// Tried and rejected (profiles/r01_k1b_forms_lds_vs_rowgroup.txt): staging the tile's
// (idx,val) stream in LDS and splitting edges evenly over lane groups -- 23.0 ms vs
// 20.3 ms at F=602; the kernel is bound by the L1/TA request path (TA_BUSY ~90 %, L2 hit
// 85 %), not by lane divergence, and the barriers + 36 KB of LDS cost occupancy.  A
// streaming form (consecutive rows per lane group as one edge stream, prefetched refills,
// 8 rows in flight) was within 2 % of this one (profiles/r01_k1b_forms_stream_vs_rowgroup.txt).
// =======================================================================================
constexpr int BLK_ROWS = 32;             // destination rows per workgroup (64: +2.5 % time, 16: same, 128: +6 %)

// nb = multiple of 8 (one block per XCD per round).  Window SB*row_bytes: measured optimum
// (profiles/r01_k1b_param_sweep.txt) is ~3.7 MB at 256-B rows and ~5 MB at 512-B rows -- a
// little over the 4 MB L2 is fine (Infinity Cache backs it), shorter (block,row) segments
// and more partial traffic are not.
uint32_t plan_blocks(uint32_t NG, uint32_t want_nb, uint32_t row_bytes, uint64_t window_bytes) {
    const uint64_t window = window_bytes ? window_bytes : row_bytes >= 512 ? (uint64_t)5242880u : (uint64_t)3932160u;
    uint32_t nb = 8;
    if (want_nb) nb = (want_nb + 7) / 8 * 8;
    else while ((uint64_t)((NG + nb - 1) / nb) * row_bytes > window && nb < (1u << 20)) nb += 8;
    return nb;
}

hipError_t build_blocked(const uint64_t *ptr, const uint32_t *idx, const float *val, uint32_t N, uint32_t NG,
                         uint64_t nnz, uint32_t want_nb, uint32_t row_bytes, BlockedAdj *out, hipStream_t s,
                         uint64_t window_bytes) {
    BlockedAdj B{};
    if (N == 0 || NG == 0) { *out = B; return hipSuccess; }
    const uint32_t nb = plan_blocks(NG, want_nb, row_bytes, window_bytes);
    B.nb = nb;
    B.SB = (NG + nb - 1) / nb;
    B.npos = N;
    B.nb_local = std::min(nb, N / B.SB);
    uint32_t *cnt = nullptr;
    hipError_t e;
#define BCK(x) if ((e = (x)) != hipSuccess) return e
    BCK(hipMalloc((void **)&cnt, (size_t)nb * N * sizeof(uint32_t)));
    BCK(hipMemsetAsync(cnt, 0, (size_t)nb * N * sizeof(uint32_t), s));
    BCK(hipMalloc((void **)&B.boff, (size_t)nb * (N + 1) * sizeof(uint32_t)));
    BCK(hipMalloc((void **)&B.bbase, (size_t)(nb + 1) * sizeof(uint64_t)));
    BCK(hipMalloc((void **)&B.bidx, (nnz ? nnz : 1) * sizeof(uint32_t)));
    BCK(hipMalloc((void **)&B.bval, (nnz ? nnz : 1) * sizeof(float)));
    uint64_t *dtotal = nullptr;
    BCK(hipMalloc((void **)&dtotal, nb * sizeof(uint64_t)));
    hipLaunchKernelGGL(blk_count_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, ptr, idx, B.SB, cnt,
                       (const uint32_t *)nullptr, (const uint16_t *)nullptr, (const uint2 *)nullptr);
    hipLaunchKernelGGL(blk_scan_kernel, dim3(nb), dim3(1024), 0, s, N, cnt, B.boff, dtotal);
    std::vector<uint64_t> tot(nb), base(nb + 1, 0);
    BCK(hipMemcpyAsync(tot.data(), dtotal, nb * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    BCK(hipStreamSynchronize(s));
    for (uint32_t b = 0; b < nb; ++b) base[b + 1] = base[b] + tot[b];
    if (base[nb] != nnz) return hipErrorUnknown;
    BCK(hipMemcpyAsync(B.bbase, base.data(), (nb + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(blk_fill_kernel, dim3((N + 255) / 256), dim3(256), 0, s, N, nb, ptr, idx, val, B.SB,
                       B.bbase, B.boff, cnt, B.bidx, B.bval, (const uint32_t *)nullptr, (const uint16_t *)nullptr,
                       (const uint2 *)nullptr);
    BCK(hipGetLastError());
    BCK(hipStreamSynchronize(s));
    (void)hipFree(cnt);
    (void)hipFree(dtotal);
    {   // hubs: (block,row) segments beyond BLK_SEG_CLAMP edges are finished by workgroup-per-chunk kernels
        std::vector<uint32_t> hoff((size_t)nb * (N + 1));
        BCK(hipMemcpy(hoff.data(), B.boff, hoff.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        std::vector<uint32_t> srow, sblk, sptr(1, 0), chunks;
        for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t *o = hoff.data() + (size_t)b * (N + 1);
            for (uint32_t v = 0; v < N; ++v) {
                const uint32_t len = o[v + 1] - o[v];
                if (len <= BLK_SEG_CLAMP) continue;
                srow.push_back(v);
                sblk.push_back(b);
                const uint64_t beg = base[b] + o[v] + BLK_SEG_CLAMP, end = base[b] + o[v + 1];
                for (uint64_t e = beg; e < end; e += BLK_SEG_CHUNK) {
                    const uint64_t e1 = e + BLK_SEG_CHUNK < end ? e + BLK_SEG_CHUNK : end;
                    chunks.insert(chunks.end(), {v, b, (uint32_t)(e & 0xFFFFFFFFu), (uint32_t)(e >> 32),
                                                 (uint32_t)(e1 & 0xFFFFFFFFu), (uint32_t)(e1 >> 32)});
                }
                sptr.push_back((uint32_t)(chunks.size() / 6));
            }
        }
        if (!srow.empty()) {
            B.seg_clamp = BLK_SEG_CLAMP;
            B.nsegs = (uint32_t)srow.size();
            B.nchunks = (uint32_t)(chunks.size() / 6);
            auto up = [&](uint32_t **dst, const std::vector<uint32_t> &h) {
                hipError_t e2 = hipMalloc((void **)dst, h.size() * sizeof(uint32_t));
                return e2 != hipSuccess ? e2 : hipMemcpy(*dst, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            };
            BCK(up(&B.seg_row, srow));
            BCK(up(&B.seg_blk, sblk));
            BCK(up(&B.seg_chunk_ptr, sptr));
            BCK(up(&B.seg_chunks, chunks));
        }
    }
#undef BCK
    *out = B;
    return hipSuccess;
}

template <int GROUP, bool UNIT, bool GH>   // GH: ids >= N are ghost rows; a partition without ghosts compiles the select out
__global__ __launch_bounds__(256) void spmm_blocked_kernel(SpmmArgs a, BlockedAdj B, float *partial,
                                                           uint32_t tiles, uint32_t round0, uint32_t rounds,
                                                           uint32_t b_lo, uint32_t b_hi) {
    constexpr int RPW = 64 / GROUP;
    constexpr int BLK_ITER = BLK_ROWS / (4 * RPW);
    const uint32_t id = blockIdx.x;
    const uint32_t xcd = id & 7u;
    uint32_t k = id >> 3;
    const uint32_t tile_seq = k % tiles;
    k /= tiles;
    const uint32_t round = round0 + k % rounds;
    const uint32_t slab = k / rounds;
    const uint32_t b = round * 8u + xcd;
    if (b < b_lo || b >= b_hi) return;   // this launch covers source blocks [b_lo, b_hi)
    // the tiles whose own rows lie in source block b go first: on a graph with locality (communities of consecutive
    // ids) they hold most of the block's edges, and starting them last would leave the XCD waiting for a few long
    // workgroups at the end of the round; on a graph without locality the rotation changes nothing
    const uint32_t tile = (uint32_t)(((uint64_t)tile_seq + (uint64_t)b * B.SB / BLK_ROWS) % tiles);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane % GROUP;
    const int gi = lane / GROUP;
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t col = slab * GROUP + li;
    const bool col_ok = col < nchunk;
    const uint32_t ccol = col_ok ? col : 0;
    const float4 *xl4 = reinterpret_cast<const float4 *>(a.xl);
    const float4 *xg4 = reinterpret_cast<const float4 *>(a.xg);
    float4 *p4 = reinterpret_cast<float4 *>(partial) + (size_t)b * a.N * nchunk;
    const uint32_t *boff = B.boff + (size_t)b * (a.N + 1);
    const uint64_t base = B.bbase[b];

#pragma unroll 1
    for (int it = 0; it < BLK_ITER; ++it) {
        const uint32_t v = tile * BLK_ROWS + (uint32_t)((it * 4 + wave) * RPW + gi);
        const bool row_ok = v < a.N;
        uint64_t e = row_ok ? base + boff[v] : 0;
        uint64_t end = row_ok ? base + boff[v + 1] : 0;
        if (B.seg_clamp && end - e > B.seg_clamp) end = e + B.seg_clamp;   // hub segment: spmm_longseg_kernel does the rest
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        while (e < end) {
            const int n = (end - e) < (uint64_t)GROUP ? (int)(end - e) : GROUP;
            uint32_t my_idx = 0;
            float my_val = 1.f;   // UNIT: unweighted sum (per-row factor applied by the reduce kernel)
            if (li < n) {
                my_idx = __builtin_nontemporal_load(B.bidx + e + li);
                if constexpr (!UNIT) my_val = __builtin_nontemporal_load(B.bval + e + li);
            }
            int j = 0;
            for (; j + 4 <= n; j += 4) {
                float4 x[4];
                float w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t s = bcast_u32<GROUP>(my_idx, j + u);
                    w[u] = UNIT ? 1.f : bcast_f32<GROUP>(my_val, j + u);
                    const float4 *row = (!GH || s < a.N) ? xl4 + (size_t)s * nchunk : xg4 + (size_t)(s - a.N) * nchunk;
                    x[u] = row[ccol];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = fma4(w[u], x[u], acc);
            }
            for (; j < n; ++j) {
                const uint32_t s = bcast_u32<GROUP>(my_idx, j);
                const float w = UNIT ? 1.f : bcast_f32<GROUP>(my_val, j);
                const float4 *row = (!GH || s < a.N) ? xl4 + (size_t)s * nchunk : xg4 + (size_t)(s - a.N) * nchunk;
                acc = fma4(w, row[ccol], acc);
            }
            e += n;
        }
        if (row_ok && col_ok) {
            v4f o = {acc.x, acc.y, acc.z, acc.w};
            __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(p4 + (size_t)v * nchunk + col));
        }
    }
}

// remainder of a long (block,row) segment: one workgroup per chunk of BLK_SEG_CHUNK edges of the blocked copy, the
// whole row width per wave, wave partials added in wave order (the K1 long-row kernel on the blocked arrays)
template <bool UNIT>
__global__ __launch_bounds__(256) void spmm_longseg_kernel(SpmmArgs a, BlockedAdj B, float *chunk_partial) {
    extern __shared__ float4 seg_lds4[];                 // [4][nchunk]
    const uint32_t *ch = B.seg_chunks + (size_t)blockIdx.x * 6;
    const uint64_t e0 = ch[2] | ((uint64_t)ch[3] << 32), e1 = ch[4] | ((uint64_t)ch[5] << 32);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nchunk = a.ld >> 2;
    const float4 *xl4 = reinterpret_cast<const float4 *>(a.xl);
    const float4 *xg4 = reinterpret_cast<const float4 *>(a.xg);
    const uint64_t q = (e1 - e0 + 3) / 4;
    const uint64_t wb = e0 + (uint64_t)wave * q, we = wb + q < e1 ? wb + q : e1;
    for (uint32_t c0 = 0; c0 < nchunk; c0 += 64) {
        const uint32_t col = c0 + lane;
        const bool act = col < nchunk;
        const uint32_t cc = act ? col : 0;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        uint64_t e = wb;
        for (; e + 4 <= we; e += 4) {
            float4 x[4];
            float w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t s = B.bidx[e + u];         // wave-uniform
                w[u] = UNIT ? 1.f : B.bval[e + u];
                x[u] = (s < a.N ? xl4 + (size_t)s * nchunk : xg4 + (size_t)(s - a.N) * nchunk)[cc];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = fma4(w[u], x[u], acc);
        }
        for (; e < we; ++e) {
            const uint32_t s = B.bidx[e];
            acc = fma4(UNIT ? 1.f : B.bval[e], (s < a.N ? xl4 + (size_t)s * nchunk : xg4 + (size_t)(s - a.N) * nchunk)[cc], acc);
        }
        if (act) seg_lds4[(size_t)wave * nchunk + col] = acc;
    }
    __syncthreads();
    float4 *p4 = reinterpret_cast<float4 *>(chunk_partial) + (size_t)blockIdx.x * nchunk;
    for (uint32_t col = threadIdx.x; col < nchunk; col += 256) {
        float4 r = seg_lds4[col];
        for (int w = 1; w < 4; ++w) {
            const float4 t = seg_lds4[(size_t)w * nchunk + col];
            r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
        }
        p4[col] = r;
    }
}

// partial[block][row,:] += the segment's chunk sums, in chunk order (one workgroup per long segment)
__global__ __launch_bounds__(256) void spmm_longseg_reduce_kernel(SpmmArgs a, BlockedAdj B, float *partial,
                                                                  const float *chunk_partial) {
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t v = B.seg_row[blockIdx.x], b = B.seg_blk[blockIdx.x];
    const uint32_t c_beg = B.seg_chunk_ptr[blockIdx.x], c_end = B.seg_chunk_ptr[blockIdx.x + 1];
    const float4 *cp4 = reinterpret_cast<const float4 *>(chunk_partial);
    float4 *row4 = reinterpret_cast<float4 *>(partial) + ((size_t)b * a.N + v) * nchunk;
    for (uint32_t col = threadIdx.x; col < nchunk; col += 256) {
        float4 acc = row4[col];
        for (uint32_t c = c_beg; c < c_end; ++c) {
            const float4 t = cp4[(size_t)c * nchunk + col];
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        row4[col] = acc;
    }
}

hipError_t launch_spmm_blocked_long_segments(const SpmmArgs &a, const BlockedAdj &B, float *partial, bool unit,
                                             float *chunk_partial, hipStream_t s) {
    if (!B.nchunks || a.ld == 0) return hipSuccess;
    const size_t lds = (size_t)4 * (a.ld >> 2) * sizeof(float4);
    if (unit) hipLaunchKernelGGL(spmm_longseg_kernel<true>, dim3(B.nchunks), dim3(256), lds, s, a, B, chunk_partial);
    else hipLaunchKernelGGL(spmm_longseg_kernel<false>, dim3(B.nchunks), dim3(256), lds, s, a, B, chunk_partial);
    hipLaunchKernelGGL(spmm_longseg_reduce_kernel, dim3(B.nsegs), dim3(256), 0, s, a, B, partial, chunk_partial);
    return hipGetLastError();
}

// out[v,:] = self[v]*xl[v,:] + sum_b partial[b][v,:]   (block order; float4 streams)
__global__ __launch_bounds__(256) void spmm_reduce_kernel(SpmmArgs a, uint32_t nb, const float *partial,
                                                          const float *row_scale) {
    const uint32_t nchunk = a.ld >> 2;
    const size_t n = (size_t)a.N * nchunk;
    const float4 *p4 = reinterpret_cast<const float4 *>(partial);
    const float4 *xl4 = reinterpret_cast<const float4 *>(a.xl);
    float4 *out4 = reinterpret_cast<float4 *>(a.out);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t v = (uint32_t)(i / nchunk);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row_scale) {   // out = self + row_scale[v] * (unweighted neighbour sum)
            for (uint32_t b = 0; b < nb; ++b) {
                const v4f p = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p4 + (size_t)b * n + i));
                acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
            }
            const float rs = row_scale[v];
            acc = make_float4(acc.x * rs, acc.y * rs, acc.z * rs, acc.w * rs);
            if (a.self_mode != 0) {
                const float sc = a.self_mode == 1 ? a.self_scale[v] : 1.f;
                const float4 x = xl4[i];
                acc.x = fmaf(x.x, sc, acc.x); acc.y = fmaf(x.y, sc, acc.y);
                acc.z = fmaf(x.z, sc, acc.z); acc.w = fmaf(x.w, sc, acc.w);
            }
        } else {
            if (a.self_mode != 0) {
                const float sc = a.self_mode == 1 ? a.self_scale[v] : 1.f;
                const float4 x = xl4[i];
                acc = make_float4(x.x * sc, x.y * sc, x.z * sc, x.w * sc);
            }
            for (uint32_t b = 0; b < nb; ++b) {
                const v4f p = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p4 + (size_t)b * n + i));
                acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
            }
        }
        if (a.accumulate) {
            const float4 p = out4[i];
            acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        }
        out4[i] = acc;
    }
}

size_t blocked_partial_bytes(const SpmmArgs &a, const BlockedAdj &B) {
    return (size_t)B.nb * a.N * a.ld * sizeof(float);
}

// partial sums of source blocks [b_lo, b_hi) (all slabs, all row tiles)
hipError_t launch_spmm_blocked_part(const SpmmArgs &a, const BlockedAdj &B, float *partial, int group, bool unit,
                                    uint32_t b_lo, uint32_t b_hi, hipStream_t s) {
    if (a.N == 0 || a.ld == 0 || b_lo >= b_hi) return hipSuccess;
    if ((a.ld & 3) || B.nb == 0 || b_hi > B.nb || (group != 8 && group != 16 && group != 32)) return hipErrorInvalidValue;
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t slabs = (nchunk + group - 1) / group;
    const uint32_t tiles = (a.N + BLK_ROWS - 1) / BLK_ROWS;
    const uint32_t round0 = b_lo / 8;
    const uint32_t rounds = (b_hi + 7) / 8 - round0;
    const uint64_t grid = (uint64_t)slabs * rounds * tiles * 8;
    if (grid > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const dim3 gr((uint32_t)grid), bl(256);
    const bool gh = a.xg != nullptr;   // callers pass nullptr when the partition has no ghost rows
#define LAUNCH_BLK_U(G, U)                                                                                    \
    do {                                                                                                      \
        if (gh)                                                                                               \
            hipLaunchKernelGGL((spmm_blocked_kernel<G, U, true>), gr, bl, 0, s, a, B, partial, tiles, round0,  \
                               rounds, b_lo, b_hi);                                                           \
        else                                                                                                  \
            hipLaunchKernelGGL((spmm_blocked_kernel<G, U, false>), gr, bl, 0, s, a, B, partial, tiles, round0, \
                               rounds, b_lo, b_hi);                                                           \
    } while (0)
#define LAUNCH_BLK(G)                                                                                         \
    do {                                                                                                      \
        if (unit) LAUNCH_BLK_U(G, true);                                                                      \
        else LAUNCH_BLK_U(G, false);                                                                          \
    } while (0)
    if (group == 8) LAUNCH_BLK(8);
    else if (group == 16) LAUNCH_BLK(16);
    else LAUNCH_BLK(32);
#undef LAUNCH_BLK
#undef LAUNCH_BLK_U
    return hipGetLastError();
}

// out = self + (row_scale *) sum_b partial[b]
hipError_t launch_spmm_blocked_reduce(const SpmmArgs &a, const BlockedAdj &B, const float *partial,
                                      const float *row_scale, hipStream_t s) {
    if (a.N == 0 || a.ld == 0) return hipSuccess;
    const uint32_t nchunk = a.ld >> 2;
    const size_t n = (size_t)a.N * nchunk;
    int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(spmm_reduce_kernel, dim3(blocks), dim3(256), 0, s, a, B.nb, partial, row_scale);
    return hipGetLastError();
}

hipError_t launch_spmm_blocked(const SpmmArgs &a, const BlockedAdj &B, float *partial, int group,
                               const float *row_scale, hipStream_t s) {
    hipError_t e = launch_spmm_blocked_part(a, B, partial, group, row_scale != nullptr, 0, B.nb, s);
    if (e != hipSuccess) return e;
    return launch_spmm_blocked_reduce(a, B, partial, row_scale, s);
}


}  // namespace dory
