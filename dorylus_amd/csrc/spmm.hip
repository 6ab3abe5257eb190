// spmm.hip -- K1 (row gather, first half of this file) and K1s (the register-accumulating gated sweep, second half: the
// default where it applies): CSC/CSR SpMM with fused self term, fp32, for gfx950.  K1b, the partial-row form between the
// two, lives in spmm_blocked.hip; the edge-regrouping kernels both blocked layouts use are spmm_common.hpp.
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
#include <algorithm>
#include <cmath>
#include <vector>

#include "sweep_core.hpp"

namespace dory {

template <int GROUP, int CHUNKS>
__global__ __launch_bounds__(256) void spmm_rows_kernel(SpmmArgs a) {
    constexpr int RPW = 64 / GROUP;      // rows per wave
    constexpr int RPB = 4 * RPW;         // rows per 256-thread block
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int li = lane % GROUP;
    const int gi = lane / GROUP;
    const uint32_t rid = blockIdx.x * RPB + wave * RPW + gi;
    const bool row_ok = rid < (a.rows ? a.rows : a.N);   // a.rows: a prefix of the row schedule (interior / boundary split)
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
    if (a.accumulate == 2) {   // second pass of an edge-split aggregation: go on from the first pass's sum (same order of additions as one pass)
        const float4 *o4 = reinterpret_cast<const float4 *>(a.out);
#pragma unroll
        for (int k = 0; k < CHUNKS; ++k)
            if (act[k]) acc[k] = o4[(size_t)v * nchunk + col[k]];
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
    uint64_t end = row_ok ? (a.ptr_end ? a.ptr_end[v] : a.ptr[v + 1]) : 0;
    if (a.row_clamp && end - e > a.row_clamp) end = e + a.row_clamp;   // the rest of a long row: spmm_longrow_kernel
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
            if (a.accumulate == 1) {
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
    const uint32_t rows = a.rows ? a.rows : a.N;
    dim3 grid((rows + RPB - 1) / RPB, (nchunk + GROUP * CHUNKS - 1) / (GROUP * CHUNKS));
    if (rows == 0 || nchunk == 0) return hipSuccess;
    hipLaunchKernelGGL((spmm_rows_kernel<GROUP, CHUNKS>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- long rows ---------------------------------------------------------------------------------------
// K1 walks a row's edge list on one lane group: a hub (a vertex with 1 % of all edges costs 300 ms at Reddit scale)
// would serialise the launch.  Rows longer than LONG_ROW_CLAMP edges are therefore cut: K1 does the self term and
// the first LONG_ROW_CLAMP edges, the remainder is split into chunks of LONG_ROW_CHUNK edges, one workgroup each
// (4 waves x a contiguous quarter of the chunk, whole row width per wave, wave partials added in wave order through
// LDS), and a last kernel adds a row's chunk sums to `out` in chunk order -- deterministic, no atomics.
struct LongChunk { uint32_t row; uint32_t pad; uint64_t e0, e1; };

__global__ __launch_bounds__(256) void spmm_longrow_kernel(SpmmArgs a, const LongChunk *chunks, float *partial) {
    extern __shared__ float4 lds4[];                    // [4][nchunk]
    const LongChunk ch = chunks[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nchunk = a.ld >> 2;
    const float4 *xl4 = reinterpret_cast<const float4 *>(a.xl);
    const float4 *xg4 = reinterpret_cast<const float4 *>(a.xg);
    const uint64_t len = ch.e1 - ch.e0, q = (len + 3) / 4;
    const uint64_t wb = ch.e0 + (uint64_t)wave * q, we = wb + q < ch.e1 ? wb + q : ch.e1;
    for (uint32_t c0 = 0; c0 < nchunk; c0 += 64) {      // 256 floats of the row per pass
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
                const uint32_t s = a.idx[e + u];        // wave-uniform
                w[u] = a.val[e + u];
                x[u] = (s < a.N ? xl4 + (size_t)s * nchunk : xg4 + (size_t)(s - a.N) * nchunk)[cc];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = fma4(w[u], x[u], acc);
        }
        for (; e < we; ++e) {
            const uint32_t s = a.idx[e];
            acc = fma4(a.val[e], (s < a.N ? xl4 + (size_t)s * nchunk : xg4 + (size_t)(s - a.N) * nchunk)[cc], acc);
        }
        if (act) lds4[(size_t)wave * nchunk + col] = acc;
    }
    __syncthreads();
    float4 *p4 = reinterpret_cast<float4 *>(partial) + (size_t)blockIdx.x * nchunk;
    for (uint32_t col = threadIdx.x; col < nchunk; col += 256) {
        float4 r = lds4[col];
        for (int w = 1; w < 4; ++w) {
            const float4 t = lds4[(size_t)w * nchunk + col];
            r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
        }
        p4[col] = r;
    }
}

// out[row,:] += sum of the row's chunk partials, in chunk order (one workgroup per long row)
__global__ __launch_bounds__(256) void spmm_longrow_reduce_kernel(SpmmArgs a, const uint32_t *row_ids,
                                                                  const uint32_t *row_chunk_ptr, const float *partial) {
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t r = row_ids[blockIdx.x], c_beg = row_chunk_ptr[blockIdx.x], c_end = row_chunk_ptr[blockIdx.x + 1];
    const float4 *p4 = reinterpret_cast<const float4 *>(partial);
    float4 *out4 = reinterpret_cast<float4 *>(a.out) + (size_t)r * nchunk;
    for (uint32_t col = threadIdx.x; col < nchunk; col += 256) {
        float4 acc = out4[col];
        for (uint32_t c = c_beg; c < c_end; ++c) {
            const float4 t = p4[(size_t)c * nchunk + col];
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        out4[col] = acc;
    }
}

void plan_long_rows(const uint64_t *ptr, uint32_t N, LongRowsHost *out) {
    out->rows.clear(); out->row_chunk_ptr.assign(1, 0); out->chunks.clear();
    for (uint32_t v = 0; v < N; ++v) {
        const uint64_t beg = ptr[v], end = ptr[v + 1];
        if (end - beg <= LONG_ROW_CLAMP) continue;
        out->rows.push_back(v);
        for (uint64_t e = beg + LONG_ROW_CLAMP; e < end; e += LONG_ROW_CHUNK) {
            out->chunks.push_back(v);                                  // row
            out->chunks.push_back(0);
            const uint64_t e1 = e + LONG_ROW_CHUNK < end ? e + LONG_ROW_CHUNK : end;
            out->chunks.push_back((uint32_t)(e & 0xFFFFFFFFu)); out->chunks.push_back((uint32_t)(e >> 32));
            out->chunks.push_back((uint32_t)(e1 & 0xFFFFFFFFu)); out->chunks.push_back((uint32_t)(e1 >> 32));
        }
        out->row_chunk_ptr.push_back((uint32_t)(out->chunks.size() / 6));
    }
}

hipError_t launch_spmm_long_rows(const SpmmArgs &a, const LongRowsDev &L, float *partial, hipStream_t s) {
    if (L.nchunks == 0 || a.ld == 0) return hipSuccess;
    const uint32_t nchunk = a.ld >> 2;
    hipLaunchKernelGGL(spmm_longrow_kernel, dim3(L.nchunks), dim3(256), (size_t)4 * nchunk * sizeof(float4), s, a,
                       reinterpret_cast<const LongChunk *>(L.chunks), partial);
    hipLaunchKernelGGL(spmm_longrow_reduce_kernel, dim3(L.nrows), dim3(256), 0, s, a, L.rows, L.row_chunk_ptr, partial);
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
    if (ch <= 96) return launch_t<32, 3>(a, s);   // e.g. 300 -> 320 floats (Amazon): 80 of 96 lanes-chunks busy instead of 80 of 128
    if (ch <= 128) return launch_t<64, 2>(a, s);
    if (ch <= 192) return launch_t<64, 3>(a, s);
    return launch_t<64, 4>(a, s);           // wider rows: gridDim.y slabs of 1024 floats
}

void free_blocked(BlockedAdj *B) {
    if (B->boff) (void)hipFree(B->boff);
    if (B->bbase) (void)hipFree(B->bbase);
    if (B->bidx) (void)hipFree(B->bidx);
    if (B->bval) (void)hipFree(B->bval);
    if (B->bent) (void)hipFree(B->bent);
    for (uint32_t *q : {B->seg_row, B->seg_blk, B->seg_chunk_ptr, B->seg_chunks, B->perm, B->otgt, B->split_rows})
        if (q) (void)hipFree(q);
    *B = BlockedAdj{};
}


// =======================================================================================
// K1s: "sweep" -- K1b's blocked adjacency (spmm_blocked.hip) without the partial rows.
//
// K1b launches one short workgroup per (tile, source block), so a partial row slab is written per block and a second
// kernel adds them up (27 GB of extra traffic per F=602 launch at Reddit scale, ~5 of 19 ms), and its windows have to be
// ~5 MB to keep the (block,row) segments long.  What the hardware can do with an L2-resident window is 30-32 TB/s
// (tools/probes/gather_probe.hip: any lanes/row, any depth; 27.5 at 4 MB, 24.8 at 5 MB, 14.9 at 8 MB).  K1s keeps the
// sums in registers instead:
//   * the destination rows are split over the XCDs (workgroup id -> XCD id & 7, for speed only); one 1024-thread
//     workgroup per CU owns NGRP*R consecutive rows of one feature slab (R rows per lane group, float4 accumulators)
//     and sweeps ALL source blocks b = 0..nb-1 in order; out = self + sum is written once, no partial buffer, no reduce;
//   * all workgroups that run on an XCD at the same time must be at (nearly) the same block for the window to be
//     L2-resident.  They are not by themselves (measured: unsynchronised sweeps gather at the L2-miss rate, 7.5 TB/s),
//     so a "sweep" = the G workgroups that are resident on an XCD together (G = CUs per XCD) is kept in step by a cheap
//     gate: a workgroup starts step b when every workgroup of its sweep has finished step b-2 (two windows live).  An
//     arrival is one L2-local atomic add into the workgroup's own word of a 128-B line per (sweep, step); the first
//     wave of a workgroup to reach a gate polls that line (8 x 16-B loads that bypass L1), the others watch LDS.
//     ~1.1 us per step (tools/probes/sync_probe.hip).  Polling is bounded and placement is only assumed, never
//     required: if the counters do not fill (workgroups placed otherwise, fewer resident than assumed) one timeout
//     switches the gates off for the launch and the kernel finishes at the unsynchronised rate -- same results;
//   * per step and lane group the R rows' segments are contiguous in the blocked copy: one coalesced load stages their
//     (idx,val) pairs in LDS (the next step's offsets and entries are already in flight across the gate), then every row
//     runs full batches of 4 gathers (nothing predicated) and one predicated tail.
// More waves per CU do not help (round 4, profiles/r04_k1s_wg_threads_experiment.patch): two 640-thread workgroups of 96
// registers are never co-resident on a CU (10 waves land 3+2+3+2 on the SIMDs; tools/probes/residency_probe.hip), five
// 256-thread workgroups are (20 waves, 8 rows per lane group) and run the launch in 12.1-13.9 ms against 12.2; twelve
// waves of 168 registers with 20 rows (two sweeps instead of three) need 15.5 ms.
// Summation order: block order, edge order inside a block -- exactly K1b's, so results agree with K1b to the last
// reassociation.  A partition with ghost rows always runs as two launches (local-source blocks, then the blocks that
// contain ghost rows with out += sum) so that the schedule that overlaps the halo exchange and the sequential one
// are the same arithmetic.  Measured at Reddit scale (tools/probes/spmm_lab.hip): F=602 14.2 ms (K1b 19.1), F=128
// 2.9 ms (3.9).  Hub (block,row) segments (B.nchunks != 0) keep K1b.
// =======================================================================================
// The skeleton (gates, loader wave, staging, batches of gathers) is sweep_core.hpp; the plain SpMM is this OP on it.
template <bool UNIT>
struct SweepPlainOp {
    static constexpr bool PLAIN = true, UNIT_W = UNIT, PROLOGUE = false, AUX_BATCH = false;
#ifndef K1S_BATCH
#define K1S_BATCH SWEEP_U   // 3 / 4 / 5 / 6 gathers per batch = 18.60 / 18.10 / 18.64 / 19.58 ms per epoch (round 5, re-measured on the final kernel)
#endif
    static constexpr int BATCH = K1S_BATCH;     // gathers per batch
    static constexpr int SLACK = SWEEP_SLACK; // windows a workgroup may run ahead of its sweep's slowest
    const float *row_scale;
    struct Row { float4 acc; };
    struct RowC {};
    typedef uint32_t Aux;                       // the entry's weight (bits)
    __device__ __forceinline__ Aux aux(uint32_t, uint32_t vbits, bool) const { return vbits; }
    template <int NB> __device__ __forceinline__ void aux_batch(const uint2 *, uint32_t, Aux (&)[NB]) const {}
    __device__ __forceinline__ void init(Row &r) const { r.acc = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ RowC row_const(uint32_t) const { return RowC{}; }
    __device__ __forceinline__ void prologue(const SpmmArgs &, const BlockedAdj &, uint32_t, uint32_t, bool, uint32_t, int) {}
    template <bool FULL>
    __device__ __forceinline__ void entry(Row &r, const RowC &, const float4 &x, uint32_t vbits, bool on) const {
        const float wv = UNIT ? 1.f : __uint_as_float(vbits);
        r.acc = fma4(FULL ? wv : (on ? wv : 0.f), x, r.acc);
    }
    // one store path for rows and for pieces of split rows (a piece: the bare sum into its slot, no scale, no self
    // term; spmm_sweep_combine_kernel finishes those rows)
    __device__ __forceinline__ void store(const Row &r, const SpmmArgs &a, const SweepArgs &w, uint32_t v, bool piece, uint32_t slot,
                                          uint32_t col, uint32_t nchunk, const float4 *xl4 /* a.xl + this lane's column */) const {
        float4 *out4 = reinterpret_cast<float4 *>(a.out);
        float4 *part4 = reinterpret_cast<float4 *>(w.split_partial);
        float4 *q = piece ? part4 + (size_t)slot * nchunk + col : out4 + (size_t)v * nchunk + col;
        const float rs = (row_scale && !piece) ? row_scale[v] : 1.f;
        float sc = 0.f;
        float4 xs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.self_mode != 0 && !piece) {
            sc = a.self_mode == 1 ? a.self_scale[v] : 1.f;
            xs = xl4[(size_t)v * nchunk];
        }
        float4 o4 = make_float4(r.acc.x * rs, r.acc.y * rs, r.acc.z * rs, r.acc.w * rs);
        o4 = fma4(sc, xs, o4);
        if (piece ? (w.flags & 2u) != 0 : a.accumulate != 0) {   // second launch / caller's out += result
            const float4 p = *q;
            o4.x += p.x; o4.y += p.y; o4.z += p.z; o4.w += p.w;
        }
        *q = o4;
    }
};

template <int GROUP, int R, bool UNIT, bool PAIR, bool LOADER>
__global__ __launch_bounds__(SWEEP_NT) void spmm_sweep_kernel(SpmmArgs a, BlockedAdj B, const float *row_scale,
                                                              SweepArgs w) {
    SweepPlainOp<UNIT> op{row_scale};
    sweep_run<GROUP, R, PAIR, LOADER>(a, B, w, op);
}

// out[row] (+)= self + (row_scale *) sum of the row's pieces, in piece order.  TPR threads per split row (a float4
// column each), several rows per workgroup on narrow tensors; eight pieces are requested at a time so that the loads of
// a hub row's hundreds of pieces overlap -- the adds stay in piece order.
__global__ __launch_bounds__(320) void spmm_sweep_combine_kernel(SpmmArgs a, BlockedAdj B, const float *row_scale,
                                                                 const float *split_partial, uint32_t tpr) {
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t rpb = blockDim.x / tpr;
    const uint32_t sr = blockIdx.x * rpb + threadIdx.x / tpr;
    if (sr >= B.nsplit) return;
    const uint32_t v = B.split_rows[3 * sr], s0 = B.split_rows[3 * sr + 1], K = B.split_rows[3 * sr + 2];
    const float4 *p4 = reinterpret_cast<const float4 *>(split_partial) + (size_t)s0 * nchunk;
    const float4 *xl4 = reinterpret_cast<const float4 *>(a.xl);
    float4 *out4 = reinterpret_cast<float4 *>(a.out) + (size_t)v * nchunk;
    for (uint32_t col = threadIdx.x % tpr; col < nchunk; col += tpr) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        uint32_t k = 0;
        for (; k + 8 <= K; k += 8) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = p4[(size_t)(k + u) * nchunk + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w; }
        }
        for (; k < K; ++k) {
            const float4 t = p4[(size_t)k * nchunk + col];
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
        if (row_scale) {
            const float rs = row_scale[v];
            acc = make_float4(acc.x * rs, acc.y * rs, acc.z * rs, acc.w * rs);
        }
        if (a.self_mode != 0) {
            const float sc = a.self_mode == 1 ? a.self_scale[v] : 1.f;
            acc = fma4(sc, xl4[(size_t)v * nchunk + col], acc);
        }
        if (a.accumulate) {
            const float4 p = out4[col];
            acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        }
        out4[col] = acc;
    }
}

hipError_t launch_spmm_sweep_combine(const SpmmArgs &a, const BlockedAdj &B, const float *row_scale, const float *split_partial,
                                     hipStream_t s) {
    if (!B.nsplit || a.ld == 0) return hipSuccess;
    const uint32_t nchunk = a.ld >> 2;
    const uint32_t tpr = std::min<uint32_t>(256u, (nchunk + 31u) & ~31u);      // threads per row: whole half-waves
    const uint32_t rpb = std::max<uint32_t>(1u, 320u / tpr);
    hipLaunchKernelGGL(spmm_sweep_combine_kernel, dim3((B.nsplit + rpb - 1) / rpb), dim3(tpr * rpb), 0, s, a, B, row_scale,
                       split_partial, tpr);
    return hipGetLastError();
}

// ---- the K1s layout ------------------------------------------------------------------------------------------------
// K1s keeps all workgroups of an XCD in step, so a step costs what its slowest lane group needs.  On a graph without
// structure the per-(block,row) segments are Poisson and the groups are even; with skewed degrees or with locality
// (communities of consecutive ids: a row's edges sit in one block) they are not -- measured with K1b's layout:
// power-law 50.9 ms against K1b's 28.9.  The layout K1s sweeps is therefore made even by construction:
//   * source rows are spread over the blocks by a seeded random permutation, local rows over blocks [0, nb_local) and
//     ghost rows over the rest (so the blocks that need no ghosts still come first); a window is then a scattered set
//     of rows -- the L2 does not care;
//   * destination rows are sorted by degree and dealt into positions, R bands of descending degree, alternate bands
//     in reverse (serpentine), so that every R consecutive positions (one lane group) carry nearly the same number of
//     edges.  A row whose degree is more than SWEEP_SPLIT x the mean cannot be evened out that way (and a hub would
//     hold up every workgroup of its XCD at every gate): it is cut into pieces of about that many edges (contiguous
//     ranges of its edge list), each piece is a position of its own whose bare sum lands in a slot of a small side
//     buffer, and spmm_sweep_combine_kernel adds a row's pieces in piece order (deterministic), scales, adds the
//     self term and writes the row.
constexpr uint32_t SWEEP_SPLIT = 2;

hipError_t build_blocked_sweep(const uint64_t *ptr, const uint32_t *idx, const float *val, uint32_t N, uint32_t NG,
                               uint64_t nnz, uint32_t want_nb, uint32_t row_bytes, uint64_t window_bytes, int R,
                               BlockedAdj *out, hipStream_t s, uint32_t layout, uint32_t sweep_tiles, uint32_t loader_relief) {
    BlockedAdj B{};
    if (N == 0 || NG == 0) { *out = B; return hipSuccess; }
    hipError_t e;
#define BCK(x) if ((e = (x)) != hipSuccess) return e
    std::vector<uint64_t> hptr((size_t)N + 1);
    BCK(hipMemcpyAsync(hptr.data(), ptr, ((size_t)N + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    BCK(hipStreamSynchronize(s));
    // destination side: pieces of rows, dealt by their edge count
    const uint64_t split_deg = std::max<uint64_t>(64, (uint64_t)SWEEP_SPLIT * (nnz / N + 1));
    struct Item { uint32_t row, k, K; uint64_t w; };
    std::vector<Item> items;
    items.reserve(N);
    std::vector<uint32_t> split_rows;       // (row, first slot, K) per split row
    uint32_t nslots = 0;
    for (uint32_t v = 0; v < N; ++v) {
        const uint64_t deg = hptr[v + 1] - hptr[v];
        if (deg > split_deg) {
            const uint64_t K64 = (deg + split_deg - 1) / split_deg;
            if (K64 > 0x7FFFFFFFull || (uint64_t)nslots + K64 > 0x7FFFFFFFull) return hipErrorInvalidValue;
            const uint32_t K = (uint32_t)K64;
            split_rows.insert(split_rows.end(), {v, nslots, K});
            for (uint32_t k = 0; k < K; ++k) items.push_back({v, k, K, (deg + K - 1) / K});
            nslots += K;
        } else {
            items.push_back({v, 0, 1, deg});
        }
    }
    if (layout & 2u) std::stable_sort(items.begin(), items.end(), [](const Item &x, const Item &y) { return x.w > y.w; });
    const uint64_t ni64 = items.size();
    if (ni64 + 16 > 0xFFFFFFF0ull) return hipErrorInvalidValue;
    const uint32_t nl = (uint32_t)ni64;
    // lane groups and their row counts, position of every item: host/sweep_deal.cpp (whole sweeps, the last sweep's
    // groups carry fewer rows; bands of descending degree, serpentine)
    uint32_t npos = std::max<uint32_t>((nl + R - 1) / (uint32_t)R * (uint32_t)R, 8);
    std::vector<uint32_t> ipos(nl);
    if (layout & 2u) {
        std::vector<uint32_t> cap;
        if (!sweep_deal_plan(nl, (uint32_t)R, sweep_tiles, &cap, &npos, loader_relief)) return hipErrorInvalidValue;
        if (loader_relief) {   // groups of different capacities on purpose: deal by weight, edges in proportion to the rows
            std::vector<uint64_t> wts(nl);
            for (uint32_t i = 0; i < nl; ++i) wts[i] = items[i].w;
            if (!sweep_deal_balanced(nl, (uint32_t)R, cap, wts.data(), ipos.data())) return hipErrorInvalidValue;
        } else if (!sweep_deal_positions(nl, (uint32_t)R, cap, ipos.data())) {
            return hipErrorInvalidValue;
        }
    } else {
        for (uint32_t i = 0; i < nl; ++i) ipos[i] = i;
    }
    std::vector<uint32_t> perm(npos, 0xFFFFFFFFu), otgt;
    std::vector<uint2> slice;
    if (nslots) { otgt.assign(npos, 0xFFFFFFFFu); slice.assign(npos, make_uint2(0u, 1u)); }
    {
        std::vector<uint32_t> slot0(nslots ? N : 0, 0);
        for (size_t q = 0; q < split_rows.size(); q += 3) slot0[split_rows[q]] = split_rows[q + 1];
        for (uint32_t i = 0; i < nl; ++i) {
            const size_t pos = ipos[i];
            perm[pos] = items[i].row;
            if (nslots) {
                slice[pos] = make_uint2(items[i].k, items[i].K);
                otgt[pos] = items[i].K > 1 ? (0x80000000u | (slot0[items[i].row] + items[i].k)) : items[i].row;
            }
        }
    }
    // source side: block of every source row
    const uint32_t G = NG - N;
    const uint64_t window = window_bytes ? window_bytes : (uint64_t)2432 << 10;
    uint32_t nbL = (uint32_t)(((uint64_t)N * row_bytes + window - 1) / window), nbG = G ? (uint32_t)(((uint64_t)G * row_bytes + window - 1) / window) : 0;
    if (want_nb) {   // explicit block count (tests): split it in proportion, at least one block each
        nbG = G ? std::max<uint32_t>(1, (uint32_t)((uint64_t)want_nb * G / NG)) : 0;
        nbL = std::max<uint32_t>(1, want_nb > nbG ? want_nb - nbG : 1);
    }
    nbL = std::max<uint32_t>(nbL, 1);
    const uint32_t nb = nbL + nbG;
    if (nb > 65535) return hipErrorInvalidValue;
    std::vector<uint16_t> sblk(NG);
    {
        uint64_t st = 0x9E3779B97F4A7C15ull ^ ((uint64_t)N << 20) ^ nnz;    // seeded: the layout is a pure function of the graph
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
        auto spread = [&](uint32_t first, uint32_t count, uint32_t b0, uint32_t nblk) {
            std::vector<uint32_t> pos(count);
            for (uint32_t i = 0; i < count; ++i) pos[i] = i;
            if (layout & 1u)
                for (uint32_t i = count; i > 1; --i) std::swap(pos[i - 1], pos[rnd() % i]);   // Fisher-Yates
            const uint32_t per = (count + nblk - 1) / nblk;
            for (uint32_t i = 0; i < count; ++i) sblk[first + i] = (uint16_t)(b0 + pos[i] / per);
        };
        spread(0, N, 0, nbL);
        if (G) spread(N, G, nbL, nbG);
    }
    B.nb = nb;
    B.SB = (std::max(N, G) + std::max(nbL, std::max<uint32_t>(nbG, 1)) - 1) / std::max(nbL, std::max<uint32_t>(nbG, 1));   // rows per block (nominal)
    B.npos = npos;
    B.nb_local = nbL;
    B.nghost = G;
    B.rows_per_group = (uint32_t)R;
    B.row_bytes = row_bytes;
    uint32_t *cnt = nullptr;
    uint16_t *d_sblk = nullptr;
    uint64_t *dtotal = nullptr;
    BCK(hipMalloc((void **)&B.perm, (size_t)npos * sizeof(uint32_t)));
    BCK(hipMemcpyAsync(B.perm, perm.data(), (size_t)npos * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    BCK(hipMalloc((void **)&d_sblk, (size_t)NG * sizeof(uint16_t)));
    BCK(hipMemcpyAsync(d_sblk, sblk.data(), (size_t)NG * sizeof(uint16_t), hipMemcpyHostToDevice, s));
    uint2 *d_slice = nullptr;
    if (nslots) {
        BCK(hipMalloc((void **)&B.otgt, (size_t)npos * sizeof(uint32_t)));
        BCK(hipMemcpyAsync(B.otgt, otgt.data(), (size_t)npos * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        BCK(hipMalloc((void **)&d_slice, (size_t)npos * sizeof(uint2)));
        BCK(hipMemcpyAsync(d_slice, slice.data(), (size_t)npos * sizeof(uint2), hipMemcpyHostToDevice, s));
        BCK(hipMalloc((void **)&B.split_rows, split_rows.size() * sizeof(uint32_t)));
        BCK(hipMemcpyAsync(B.split_rows, split_rows.data(), split_rows.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        B.nsplit = (uint32_t)(split_rows.size() / 3);
        B.nslots = nslots;
    }
    BCK(hipMalloc((void **)&cnt, (size_t)nb * npos * sizeof(uint32_t)));
    BCK(hipMemsetAsync(cnt, 0, (size_t)nb * npos * sizeof(uint32_t), s));
    BCK(hipMalloc((void **)&B.boff, (size_t)nb * (npos + 1) * sizeof(uint32_t)));
    BCK(hipMalloc((void **)&B.bbase, (size_t)(nb + 1) * sizeof(uint64_t)));
    // (the loader copies whole 1 KB runs: up to 128 entries past a lane group's last one are read, never used)
    BCK(hipMalloc((void **)&B.bent, (nnz + 130) * sizeof(uint2)));
    BCK(hipMemsetAsync(B.bent + nnz, 0, 130 * sizeof(uint2), s));
    BCK(hipMalloc((void **)&dtotal, nb * sizeof(uint64_t)));
    hipLaunchKernelGGL(blk_count_kernel, dim3((npos + 255) / 256), dim3(256), 0, s, npos, ptr, idx, B.SB, cnt, B.perm, d_sblk,
                       (const uint2 *)d_slice);
    hipLaunchKernelGGL(blk_scan_kernel, dim3(nb), dim3(1024), 0, s, npos, cnt, B.boff, dtotal);
    std::vector<uint64_t> tot(nb), base(nb + 1, 0);
    BCK(hipMemcpyAsync(tot.data(), dtotal, nb * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    BCK(hipStreamSynchronize(s));   // also: perm / sblk / heavy host vectors are free to go
    for (uint32_t b = 0; b < nb; ++b) base[b + 1] = base[b] + tot[b];
    if (base[nb] != nnz) return hipErrorUnknown;
    BCK(hipMemcpyAsync(B.bbase, base.data(), (nb + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(blk_fill_kernel, dim3((npos + 255) / 256), dim3(256), 0, s, npos, nb, ptr, idx, val, B.SB, B.bbase,
                       B.boff, cnt, (uint32_t *)nullptr, (float *)nullptr, B.perm, d_sblk, (const uint2 *)d_slice, B.bent);
    BCK(hipGetLastError());
    BCK(hipStreamSynchronize(s));
    (void)hipFree(cnt);
    (void)hipFree(dtotal);
    (void)hipFree(d_sblk);
    if (d_slice) (void)hipFree(d_slice);
#undef BCK
    *out = B;
    return hipSuccess;
}

// rows per lane group: the choice that leaves the fewest idle workgroup slots in the last sweep of a slab;
// force_r (option spmm_sweep_rows of the context; tests, experiments): 0 = pick by fill
int sweep_pick_r(uint32_t N, int group, uint32_t G, int force_r, int max_r) {
    if (force_r == 2 || force_r == 4 || force_r == 6 || force_r == 8 || (force_r == 10 && group == 32 && max_r >= 10) ||
        ((force_r == 3 || force_r == 5) && group == 16))
        return force_r;
    const uint32_t rpx = (N + 7) / 8;
    int best = 8;
    double best_fill = 0;
    for (int R : {10, 8, 6, 4, 2}) {            // few rows per group: small partitions (one of 8 ranks) still fill every CU
        if ((group == 16 && R == 10) || R > max_r) continue;   // 16-lane groups stage twice the entries per lane: 10 rows would spill
        const uint32_t RW = (uint32_t)(SWEEP_NT / group) * R;
        const uint32_t tiles = (rpx + RW - 1) / RW;
        const uint32_t spp = (tiles + G - 1) / G;
        const double fill = (double)rpx / ((double)spp * G * RW);
        if (fill > best_fill + 0.02) { best_fill = fill; best = R; }
    }
    return best;
}

bool sweep_supported(const SpmmArgs &a, const BlockedAdj &B, int group) {
    // source rows are addressed through a buffer resource: 32-bit byte offsets, row id x row bytes in 24 x 24 bits
    const uint64_t row_b = (uint64_t)a.ld * 4u;
    const bool addr_ok = (uint64_t)a.N * row_b < (1ull << 32) && (uint64_t)B.nghost * row_b < (1ull << 32) &&
                         (uint64_t)a.N + B.nghost < (1u << 24) && row_b < (1u << 24);
    return (group == 16 || group == 32) && !(a.ld & 3) && B.nb > 0 && B.nchunks == 0 && a.N >= 8 && B.npos >= 8 && addr_ok;
}

// rows per lane group of a launch: what the layout was dealt for, unless forced (option) or not instantiated for the
// lane-group width
static int sweep_rows_for(const BlockedAdj &B, int group, uint32_t G, int force_r) {
    const int forced = sweep_pick_r(0, group, G, force_r, 10);      // (returns the forced value whatever N when one is valid)
    if (force_r && forced == force_r) return forced;
    // (16-lane groups stage twice the entries per lane: eight rows spill two registers into the chain LDS -> gathers -> sums;
    // a spilling variant is not launched unless an option forces it -- tests/test_kernel_resources.py)
    // 16-lane launches (64-float rows) have twice the lane groups per workgroup: half the layout's rows per group walks the
    // rows per workgroup and step the layout was dealt for, and leaves registers for the loader wave (round 5: the 64-float
    // aggregations of the GAT prototype 2.00 -> see DESIGN; 6 rows without the loader was the round-4 form)
    if (B.rows_per_group && group == 16) return std::max<int>(2, (int)B.rows_per_group / 2);
    if (B.rows_per_group && group == 32) return (int)B.rows_per_group;
    return sweep_pick_r(B.npos, group, G, 0, 10);
}

// counter words one launch over nblocks source blocks needs (callers size the scratch for the largest launch)
size_t sweep_scratch_bytes(const BlockedAdj &B, uint32_t ld, int group, uint32_t G, uint32_t nblocks, int force_r) {
    const int R = sweep_rows_for(B, group, G, force_r);
    const uint32_t RW = (uint32_t)(SWEEP_NT / group) * R;
    const uint32_t Gmin = G > 12 ? G - 8 : G;      // launches may leave up to 8 CUs per XCD to concurrent kernels
    const uint32_t rpx = (B.npos + 7) / 8 + 16, tiles = (rpx + RW - 1) / RW + 1, spp = (tiles + Gmin - 1) / Gmin;
    const uint32_t slabs = ((ld >> 2) + group - 1) / group;
    return ((size_t)8 * slabs * spp * nblocks * 32 + 1) * sizeof(uint32_t);
}

// out (+)= self + (row_scale *) sum over source blocks [b_lo, b_hi); `done` = sweep_scratch_bytes() of device memory
hipError_t launch_spmm_sweep(const SpmmArgs &a, const BlockedAdj &B, int group, const float *row_scale, uint32_t cus,
                             uint32_t b_lo, uint32_t b_hi, uint32_t *done, hipStream_t s, const SweepCtl &ctl, uint32_t flags,
                             float *split_partial, uint32_t reserve) {
    if (a.N == 0 || a.ld == 0 || b_lo >= b_hi) return hipSuccess;
    if (!sweep_supported(a, B, group) || b_hi > B.nb || cus == 0 || cus > 32 || !ctl.stat) return hipErrorInvalidValue;
    if (b_lo < B.nb_local && b_hi > B.nb_local) return hipErrorInvalidValue;   // one source array per launch
    if (b_lo >= B.nb_local && !a.xg) return hipErrorInvalidValue;
    const int R = sweep_rows_for(B, group, cus, ctl.force_r);
    // A sweep is the workgroups that must be resident on an XCD together; each takes a whole CU (all its registers).
    // While other kernels hold CUs (the exchange's RCCL kernels under the local-source launch) fewer fit: a smaller
    // sweep leaves them room -- the surplus workgroups of the next sweep simply wait at their first gates.
    const uint32_t G = cus > 12 ? cus - std::min<uint32_t>(reserve, 8u) : cus;
    SweepArgs w{};
    const uint32_t RW = (uint32_t)(SWEEP_NT / group) * R;
    w.rpx = ((B.npos + 7) / 8 + R - 1) / R * R;   // whole lane groups per XCD
    w.tiles_x = (w.rpx + RW - 1) / RW;
    w.G = G;
    const uint32_t spp = (w.tiles_x + G - 1) / G;
    const uint32_t slabs = ((a.ld >> 2) + group - 1) / group;
    w.nsweeps = slabs * spp;
    w.b_lo = b_lo; w.b_hi = b_hi;
    w.done = done;
    w.flags = flags | (reserve ? 32u : 0u);
    w.split_partial = split_partial;
    w.stat = ctl.stat;
    if (B.nslots && !split_partial) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(done, 0, ((size_t)8 * w.nsweeps * (b_hi - b_lo) * 32 + 1) * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const dim3 gr(8u * slabs * spp * G), bl(SWEEP_NT);
    const bool unit = row_scale != nullptr;
    // rows in pairs (one stream of entries per two rows) pay on launches of several slabs (five slabs, F=602: 13.8 ->
    // 13.3 ms; four: 11.0 -> 10.8; three: 8.16 -> 8.1), not on one or two (F=128: 2.70 -> 2.78; F=256: 5.43 -> 5.46);
    // ctl.pair (option spmm_sweep_pair): -1 = that rule, 0 / 1 = forced (experiments)
    const bool pair = (R & 1) ? false : (ctl.pair < 0 ? slabs >= 3 : ctl.pair != 0);   // (odd R: 16-lane launches on a 6- or 10-row layout)
#define SWEEP_LAUNCH_L(GRP, RR, LD)                                                                                    \
    do {                                                                                                               \
        if (unit) { if (pair) hipLaunchKernelGGL((spmm_sweep_kernel<GRP, RR, true, true, LD>), gr, bl, 0, s, a, B, row_scale, w);   \
                    else hipLaunchKernelGGL((spmm_sweep_kernel<GRP, RR, true, false, LD>), gr, bl, 0, s, a, B, row_scale, w); }   \
        else { if (pair) hipLaunchKernelGGL((spmm_sweep_kernel<GRP, RR, false, true, LD>), gr, bl, 0, s, a, B, row_scale, w);       \
               else hipLaunchKernelGGL((spmm_sweep_kernel<GRP, RR, false, false, LD>), gr, bl, 0, s, a, B, row_scale, w); }       \
    } while (0)
#define SWEEP_LAUNCH(GRP, RR)                                                                                          \
    do {                                                                                                               \
        if (ctl.loader) SWEEP_LAUNCH_L(GRP, RR, true); else SWEEP_LAUNCH_L(GRP, RR, false);                            \
    } while (0)
#define SWEEP_LAUNCH_R(GRP)                                                                                            \
    do {                                                                                                               \
        if (R == 8) SWEEP_LAUNCH(GRP, 8); else if (R == 6) SWEEP_LAUNCH(GRP, 6); else if (R == 4) SWEEP_LAUNCH(GRP, 4);  \
        else SWEEP_LAUNCH(GRP, 2);                                                                                     \
    } while (0)
    if (group == 32) { if (R == 10) SWEEP_LAUNCH(32, 10); else SWEEP_LAUNCH_R(32); }
    else { if (R == 5) SWEEP_LAUNCH(16, 5); else if (R == 3) SWEEP_LAUNCH(16, 3); else SWEEP_LAUNCH_R(16); }
#undef SWEEP_LAUNCH_R
#undef SWEEP_LAUNCH
#undef SWEEP_LAUNCH_L
    return hipGetLastError();
}

// ---- diagnostic: hold CUs the way a concurrent kernel (an exchange's RCCL kernels, a co-tenant) would -----------------
// `workgroups` workgroups of 1024 threads, all 128 registers per lane (nothing else fits beside one on a CU) and 96 KB of
// LDS each, asleep for `ticks` of the 100 MHz wall clock.  K1s's sweeps then find fewer free CUs per XCD than they
// assume: tests/test_gpu_gates.py.
__global__ __launch_bounds__(1024) void occupy_cus_kernel(uint64_t ticks, uint32_t *sink) {
    extern __shared__ uint32_t occupy_lds[];
    asm volatile("v_mov_b32 v127, 0" ::: "v127");   // a 128-register allocation: 4 waves per SIMD are the whole file
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (ticks == ~0ull) { occupy_lds[threadIdx.x] = 1; sink[0] = occupy_lds[0]; }
}
hipError_t launch_occupy_cus(uint32_t workgroups, uint64_t usec, hipStream_t s) {
    if (!workgroups) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(occupy_cus_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 << 10);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(occupy_cus_kernel, dim3(workgroups), dim3(1024), 96 << 10, s, usec * 100ull, (uint32_t *)nullptr);
    return hipGetLastError();
}

// ---- which XCD does workgroup i run on?  K1s assumes i & 7 (for speed only); dory_create checks it once per context ----
__global__ void xcd_probe_kernel(uint32_t *xcc) {
    const uint32_t id = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);   // HW_REG_XCC_ID, bits [3:0]
    if (threadIdx.x == 0) xcc[blockIdx.x] = id & 15u;
}
hipError_t launch_xcd_probe(uint32_t *xcc, uint32_t grid, hipStream_t s) {
    hipLaunchKernelGGL(xcd_probe_kernel, dim3(grid), dim3(64), 0, s, xcc);
    return hipGetLastError();
}

}  // namespace dory
