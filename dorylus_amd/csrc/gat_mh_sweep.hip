// gat_mh_sweep.hip -- multi-head GAT extension (definition: gat_mh.hip; oracle: oracle/gat_mh_oracle.py): the edge passes on
// K1s's skeleton (sweep_core.hpp) -- register-resident sums over ALL source blocks, per-XCD gates, loader wave, the even
// layout of build_blocked_sweep -- instead of one short workgroup per (tile, source block) that leaves a partial row per
// block for a reduce kernel (gat_mh_blocked.hip, kept for the shapes this file does not cover).
//
// What makes the softmax fit that structure is a per-(v,k) UPPER-BOUND shift instead of the running maximum:
//     m[v,k] = LeakyReLU(max_u el[u,k] + er[v,k])  >=  s(u,v,k) = LeakyReLU(el[u,k] + er[v,k])   for every source u
// (LeakyReLU is monotone; max_u over the local and ghost source rows: one tiny reduction per layer).  With a shift that is
// known before the sweep the softmax is single-pass and LINEAR in the edges: one exp per (edge, head), no running maximum,
// no rescaling of the sums, no (m_b, den_b) merge; pieces of split rows and the two launches of a partitioned run (local-
// source blocks, ghost blocks) simply add.  alpha = exp(s - m) / den is the same number whatever m is, so the backward
// passes read (m, den) exactly as before.  The price: if the scores of a row's own neighbourhood lie far below the global
// bound (more than ~69 in natural-log units), den underflows; such rows are detected in the finishing kernel
// (den < GATMH_DEN_TINY) and recomputed by a row-wise online-softmax kernel with their true maximum (tested).
// Inside the sweep everything is in log2 units (a_l, er, m scaled by log2 e once): exp(s - m) = v_exp_f32(max(t1, t2)),
// t1 = el' + c1, t2 = 0.2 el' + c2 with c1 = er' - m', c2 = 0.2 er' - m' per (row, head) in an LDS table.
#include "gat_mh.hpp"
#include "sweep_core.hpp"

namespace dory {

constexpr float GATMH_LOG2E = 1.4426950408889634f;
constexpr float GATMH_DEN_TINY = 1e-30f;

template <int CTRL>
__device__ __forceinline__ float sw_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the HL neighbouring lanes of a head, all inside the VALU (gat_mh_blocked.hip: head_lanes_sum); HL <= 16
template <int HL>
__device__ __forceinline__ float sw_head_sum(float v) {
    if constexpr (HL >= 2) v += sw_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
    if constexpr (HL >= 4) v += sw_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
    if constexpr (HL >= 8) v += sw_dpp<0x141>(v);   // row_half_mirror
    if constexpr (HL >= 16) v += sw_dpp<0x140>(v);  // row_mirror
    return v;
}

// ---- max_u el[u,k]: ordered-int keys so that an integer atomic max orders floats -----------------------------------
__device__ __forceinline__ int gm_fkey(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__device__ __forceinline__ float gm_fkey_inv(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

__global__ __launch_bounds__(256) void gatmh_elmax_kernel(uint32_t N, uint32_t G, uint32_t K, uint32_t ldk, const float *el,
                                                          const float *elg, int *key /*[K], preset to INT_MIN-ish*/) {
    __shared__ float red[256];
    uint32_t KP = 1;
    while (KP < K) KP <<= 1;                 // K <= 64
    const uint32_t k = threadIdx.x % KP, j = threadIdx.x / KP, RPI = 256 / KP;
    float mx = -INFINITY;
    if (k < K)
        for (uint64_t r = (uint64_t)blockIdx.x * RPI + j; r < (uint64_t)N + G; r += (uint64_t)gridDim.x * RPI)
            mx = fmaxf(mx, r < N ? el[r * ldk + k] : elg[(r - N) * ldk + k]);
    red[threadIdx.x] = mx;
    __syncthreads();
    for (uint32_t s = 128; s >= KP; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x < K && red[threadIdx.x] > -INFINITY) atomicMax(key + threadIdx.x, gm_fkey(red[threadIdx.x]));
}

// ---- forward: acc[v,:] = sum_e exp(s_e - m_v) z[src(e),:],  den[v,k] = sum_e exp(s_e - m_v)  (self edge: finish kernel) ---
#ifndef GATMH_PIPE
#define GATMH_PIPE 0
#endif
template <int GROUP, int HL, int R>
struct GatFwdSweepOp {
    static constexpr bool PLAIN = false, UNIT_W = true, PROLOGUE = true;
    static constexpr int PIPE = GATMH_PIPE;   // entries per batch of the pipelined walk (sweep_core.hpp); 0 = the classic walk
    static constexpr int HPS = GROUP / HL;                     // heads per slab of GROUP lanes
    static constexpr int RW = (SWEEP_NT / GROUP) * R;
    // arguments
    const float *er, *a_l;
    const int *elmax_key;
    float *dacc;        // [N][ldk]: unnormalised denominators (the finish kernel adds the self edge and writes den)
    float *den_slots;   // [nslots][ldk]: the same for pieces of split rows
    uint32_t K, D, ldk;
    // per thread
    float4 al4;
    uint32_t k, hl;
    const float2 *ctab;
    struct Row { float4 acc; float den; };
    struct RowC { float c1, c2; };
    __device__ __forceinline__ void init(Row &r) const { r.acc = make_float4(0.f, 0.f, 0.f, 0.f); r.den = 0.f; }
    __device__ __forceinline__ void prologue(const SpmmArgs &a, const BlockedAdj &B, uint32_t pos0, uint32_t xend, uint32_t, uint32_t col, int li) {
        __shared__ float2 tab[RW * HPS];
        const uint32_t head0 = (col / GROUP) * HPS;            // first head of this slab
        for (uint32_t i = threadIdx.x; i < (uint32_t)(RW * HPS); i += SWEEP_NT) {
            const uint32_t lrow = i / HPS, kk = head0 + i % HPS, pos = pos0 + lrow;
            const uint32_t v = pos < xend ? (B.perm ? B.perm[pos] : pos) : 0xFFFFFFFFu;
            float2 c = make_float2(0.f, 0.f);
            if (v != 0xFFFFFFFFu && kk < K) {
                const float e = er[(size_t)v * ldk + kk];
                const float mm = lrelu02(gm_fkey_inv(elmax_key[kk]) + e);
                c = make_float2((e - mm) * GATMH_LOG2E, (GATMH_SLOPE * e - mm) * GATMH_LOG2E);
            }
            tab[i] = c;
        }
        ctab = tab;
        hl = (uint32_t)li / HL;
        k = min(head0 + hl, K - 1);
        const uint32_t f0 = col * 4, KD = K * D;               // a_l is a dense K x D vector (41-feature heads end mid-float4)
        al4 = make_float4(f0 < KD ? a_l[f0] * GATMH_LOG2E : 0.f, f0 + 1 < KD ? a_l[f0 + 1] * GATMH_LOG2E : 0.f,
                          f0 + 2 < KD ? a_l[f0 + 2] * GATMH_LOG2E : 0.f, f0 + 3 < KD ? a_l[f0 + 3] * GATMH_LOG2E : 0.f);
    }
    __device__ __forceinline__ RowC row_const(uint32_t lrow) const {
        const float2 c = ctab[lrow * HPS + hl];
        return RowC{c.x, c.y};
    }
    // The sweep is bound by the vector ALU as much as by the addresser (16 vector instructions per gather instruction in the
    // first cut, 3.78 ms per 128-float launch whatever the rows per group or the gates): everything here is written for
    // the packed fp32 instructions (v_pk_mul / v_pk_fma: two lanes' worth per issue slot).
    template <bool FULL>
    __device__ __forceinline__ void entry(Row &r, const RowC &c, const float4 &x, uint32_t, bool on) const {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 xlo = {x.x, x.y}, xhi = {x.z, x.w}, alo = {al4.x, al4.y}, ahi = {al4.z, al4.w};
        const f2 p = __builtin_elementwise_fma(xhi, ahi, xlo * alo);                 // v_pk_mul + v_pk_fma
        const float e = sw_head_sum<HL>(p.x + p.y);                                   // el'[src] of this lane's head
        const f2 ee = {e, e}, kk = {1.f, GATMH_SLOPE}, cc = {c.c1, c.c2};
        const f2 t = __builtin_elementwise_fma(ee, kk, cc);                            // (e + c1, 0.2 e + c2): one v_pk_fma
        float al = __builtin_amdgcn_exp2f(fmaxf(t.x, t.y));
        if constexpr (!FULL) al = on ? al : 0.f;               // (an absent slot gathered zeros: its score is not zero)
        r.den += al;
        r.acc = fma4(al, x, r.acc);
    }
    __device__ __forceinline__ void store(const Row &r, const SpmmArgs &a, const SweepArgs &w, uint32_t v, bool piece, uint32_t slot,
                                          uint32_t col, uint32_t nchunk, const float4 *) const {
        float4 *q = piece ? reinterpret_cast<float4 *>(w.split_partial) + (size_t)slot * nchunk + col
                          : reinterpret_cast<float4 *>(a.out) + (size_t)v * nchunk + col;
        float *dq = piece ? den_slots + (size_t)slot * ldk + k : dacc + (size_t)v * ldk + k;
        float4 o4 = r.acc;
        float dn = r.den;
        const bool head_lane = (threadIdx.x % HL) == 0 && col * 4 < K * D;
        if (piece ? (w.flags & 2u) != 0 : a.accumulate != 0) {   // second launch of a partitioned run (ghost blocks)
            const float4 p = *q;
            o4.x += p.x; o4.y += p.y; o4.z += p.z; o4.w += p.w;
            if (head_lane) dn += *dq;
        }
        *q = o4;
        if (head_lane) *dq = dn;
    }
};

template <int GROUP, int HL, int R, bool LOADER>
__global__ __launch_bounds__(SWEEP_NT) void gatmh_forward_sweep_kernel(SpmmArgs a, BlockedAdj B, SweepArgs w, const float *er,
                                                                       const float *a_l, const int *elmax_key, float *dacc,
                                                                       float *den_slots, uint32_t K, uint32_t D, uint32_t ldk) {
    GatFwdSweepOp<GROUP, HL, R> op{er, a_l, elmax_key, dacc, den_slots, K, D, ldk};
    sweep_run<GROUP, R, false, LOADER>(a, B, w, op);
}

// pieces of split rows: o[v,:] = sum of the pieces' slots, dacc[v,k] likewise (piece order)
__global__ __launch_bounds__(256) void gatmh_sweep_combine_kernel(BlockedAdj B, uint32_t ld, uint32_t ldk, uint32_t K, const float *part,
                                                                  const float *den_slots, float *o, float *dacc) {
    const uint32_t sr = blockIdx.x;
    if (sr >= B.nsplit) return;
    const uint32_t v = B.split_rows[3 * sr], s0 = B.split_rows[3 * sr + 1], P = B.split_rows[3 * sr + 2];
    for (uint32_t f = threadIdx.x; f < ld + K; f += blockDim.x) {
        float s = 0.f;
        if (f < ld) {
            for (uint32_t p = 0; p < P; ++p) s += part[(size_t)(s0 + p) * ld + f];
            o[(size_t)v * ld + f] = s;
        } else {
            for (uint32_t p = 0; p < P; ++p) s += den_slots[(size_t)(s0 + p) * ldk + (f - ld)];
            dacc[(size_t)v * ldk + (f - ld)] = s;
        }
    }
}

// o = (acc + e_self z_v) / (dacc + e_self); m, den for the backward passes; rows whose denominator underflowed are listed
__global__ __launch_bounds__(256) void gatmh_forward_finish_kernel(GatMhArgs a, const float *z, const float *el, const float *er,
                                                                   const int *elmax_key, const float *dacc, float *o, float *m_out,
                                                                   float *den_out, uint32_t *redo_flag, uint32_t *redo_list /*[0] = count*/) {
    const uint32_t nchunk = a.ld >> 2;
    const size_t n = (size_t)a.N * nchunk;
    const float4 *z4 = reinterpret_cast<const float4 *>(z);
    float4 *o4 = reinterpret_cast<float4 *>(o);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t v = (uint32_t)(i / nchunk), col = (uint32_t)(i % nchunk);
        if (col * 4 >= a.K * a.D) continue;
        const uint32_t k = min((col * 4) / a.D, a.K - 1);
        const size_t vk = (size_t)v * a.ldk + k;
        const float e = er[vk];
        const float mm = lrelu02(gm_fkey_inv(elmax_key[k]) + e);
        const float es = __builtin_amdgcn_exp2f((lrelu02(el[vk] + e) - mm) * GATMH_LOG2E);
        const float dn = dacc[vk] + es;
        const float idn = 1.f / dn;
        const float4 acc = o4[i], x = z4[i];
        o4[i] = make_float4(fmaf(es, x.x, acc.x) * idn, fmaf(es, x.y, acc.y) * idn, fmaf(es, x.z, acc.z) * idn, fmaf(es, x.w, acc.w) * idn);
        if ((col * 4) % a.D < 4 || a.K == 1) {
            if (a.K != 1 || col == 0) { m_out[vk] = mm; den_out[vk] = dn; }
            if (!(dn >= GATMH_DEN_TINY) && atomicExch(redo_flag + v, 1u) == 0u) redo_list[1 + atomicAdd(redo_list, 1u)] = v;
        }
    }
}

// the rows the finish kernel listed, recomputed with their own maximum (online softmax over the row's in-edges and the
// self edge, one wave per row, lanes over the features); rare by construction, so nothing here is tuned
__global__ __launch_bounds__(256) void gatmh_forward_redo_kernel(GatMhArgs a, const float *z, const float *zg, const float *el,
                                                                 const float *elg, const float *er, float *o, float *m_out, float *den_out,
                                                                 uint32_t *redo_flag, const uint32_t *redo_list) {
    const uint32_t cnt = redo_list[0];
    const int lane = threadIdx.x & 63;
    const uint32_t KD = a.K * a.D;
    for (uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6); i < cnt; i += gridDim.x * 4) {
        const uint32_t v = redo_list[1 + i];
        for (uint32_t f = (uint32_t)lane; f < KD; f += 64) {
            const uint32_t k = f / a.D;
            const float er_v = er[(size_t)v * a.ldk + k];
            float mx = -INFINITY, den = 0.f, acc = 0.f;
            const uint64_t e0 = a.ptr[v], e1 = a.ptr[v + 1];
            for (uint64_t e = e0; e <= e1; ++e) {                 // e1 stands for the self edge
                const uint32_t u = e < e1 ? a.idx[e] : v;
                const bool loc = u < a.N;
                const float el_u = loc ? el[(size_t)u * a.ldk + k] : elg[(size_t)(u - a.N) * a.ldk + k];
                const float zu = loc ? z[(size_t)u * a.ld + f] : zg[(size_t)(u - a.N) * a.ld + f];
                const float s = lrelu02(el_u + er_v);
                const float mn = fmaxf(mx, s);
                const float sc = __expf(mx - mn), al = __expf(s - mn);   // (exp(-inf) = 0 on the first edge)
                acc = fmaf(acc, sc, al * zu);
                den = fmaf(den, sc, al);
                mx = mn;
            }
            o[(size_t)v * a.ld + f] = acc / den;
            if (f % a.D == 0) { m_out[(size_t)v * a.ldk + k] = mx; den_out[(size_t)v * a.ldk + k] = den; }
        }
        if (lane == 0) redo_flag[v] = 0u;
    }
}

// ---- launchers ------------------------------------------------------------------------------------------------------
// lanes per head on a slab of `group` lanes; 0 = a shape the sweep kernels do not cover (the blocked kernels take it)
int gatmh_sweep_hl(uint32_t K, uint32_t D, uint32_t ld) {
    const int group = ld >= 128 ? 32 : 16;
    if (!gatmh_shape_ok(K, D) || (ld & 3)) return 0;
    if (K == 1) return (ld <= 64 && group == 16) ? 16 : 0;        // a single head: the whole (one-slab, 16-lane) row
    if ((D & 3) || (D & (D - 1))) return 0;
    const int hl = (int)(D / 4);
    return (hl >= 2 && hl <= 16 && hl <= group) ? hl : 0;
}

// rows per lane group of a launch.  The layout is dealt for the 32-lane launches (at most 8 rows: 123 registers, nothing
// spilled); a 16-lane launch stages twice the entries per lane and would spill from 6 rows on, and a kernel that spills is
// not an option here: with several contexts on one device (one stream each, P partitions in one process) the first
// concurrent launches of a spilling variant returned wrong sums (measured, round 5: 14 spilled registers, 5 of 6 fresh
// processes wrong, none with 4 rows or with the contexts serialised) -- so it walks the same positions 4 rows per group.
int gatmh_sweep_rows(const BlockedAdj &S, int group) {
    const int r = (int)S.rows_per_group;
    return group == 16 ? std::min(r, 4) : r;
}

// scratch layout of one layer's forward (floats): [pieces: nslots x ld][den slots: nslots x ldk][dacc: N x ldk][keys: 64]
// [redo flags: N][redo list: 1 + N]
size_t gatmh_sweep_scratch_bytes(const BlockedAdj &S, uint32_t N, uint32_t ld, uint32_t ldk) {
    return ((size_t)S.nslots * (ld + ldk) + (size_t)N * ldk + 64 + (size_t)N + 1 + (size_t)N) * sizeof(float) + 256;
}
struct GatSweepScratch {
    float *pieces, *den_slots, *dacc;
    int *keys;
    uint32_t *redo_flag, *redo_list;
};
static GatSweepScratch gatmh_carve(float *scratch, const BlockedAdj &S, uint32_t N, uint32_t ld, uint32_t ldk) {
    GatSweepScratch c;
    c.pieces = scratch;
    c.den_slots = c.pieces + (size_t)S.nslots * ld;
    c.dacc = c.den_slots + (size_t)S.nslots * ldk;
    c.keys = reinterpret_cast<int *>(c.dacc + (size_t)N * ldk);
    c.redo_flag = reinterpret_cast<uint32_t *>(c.keys + 64);
    c.redo_list = c.redo_flag + N;
    return c;
}

// the shift: keys <- max over the local and ghost rows of el.  Once per layer, before the first part launch.
hipError_t launch_gatmh_sweep_begin(uint32_t N, uint32_t G, uint32_t K, uint32_t ld, uint32_t ldk, const BlockedAdj &S, const float *el,
                                    const float *elg, float *scratch, hipStream_t s) {
    if (N == 0) return hipSuccess;
    const GatSweepScratch c = gatmh_carve(scratch, S, N, ld, ldk);
    hipError_t e = hipMemsetAsync(c.keys, 0x80, 64 * sizeof(int), s);          // 0x80808080: below the key of every float
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(c.redo_list, 0, sizeof(uint32_t), s)) != hipSuccess) return e;
    // (the scratch buffer is everybody's and the carve depends on the layer's widths: the flags are cleared every time)
    if ((e = hipMemsetAsync(c.redo_flag, 0, (size_t)N * sizeof(uint32_t), s)) != hipSuccess) return e;
    const uint64_t rows = (uint64_t)N + G;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(1024, (rows + 255) / 256 + 1);
    hipLaunchKernelGGL(gatmh_elmax_kernel, dim3(blocks), dim3(256), 0, s, N, G, K, ldk, el, elg, c.keys);
    return hipGetLastError();
}

// one launch over the source blocks [b_lo, b_hi) of the sweep layout S (a partition with ghost rows: local-source blocks
// first, then the ghost blocks with accumulate = true)
hipError_t launch_gatmh_forward_sweep_part(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const BlockedAdj &S,
                                           const float *z, const float *zg, const float *er, const float *a_l, float *o, float *scratch,
                                           uint32_t cus, uint32_t b_lo, uint32_t b_hi, bool accumulate, uint32_t *done, const SweepCtl &ctl,
                                           uint32_t flags, hipStream_t s) {
    if (N == 0 || b_lo >= b_hi) return hipSuccess;
    const int group = ld >= 128 ? 32 : 16;
    const int HL = gatmh_sweep_hl(K, D, ld);
    SpmmArgs a{};
    a.N = N; a.F = K * D; a.ld = ld; a.xl = z; a.xg = zg; a.out = o; a.accumulate = accumulate ? 1 : 0; a.self_mode = 0;
    if (!HL || !sweep_supported(a, S, group) || b_hi > S.nb || cus == 0 || cus > 32 || !ctl.stat) return hipErrorInvalidValue;
    if (b_lo < S.nb_local && b_hi > S.nb_local) return hipErrorInvalidValue;
    if (b_lo >= S.nb_local && !zg) return hipErrorInvalidValue;
    const int R = gatmh_sweep_rows(S, group);
    if (R != 8 && R != 6 && R != 4 && R != 2) return hipErrorInvalidValue;
    const GatSweepScratch c = gatmh_carve(scratch, S, N, ld, ldk);
    SweepArgs w{};
    const uint32_t RW = (uint32_t)(SWEEP_NT / group) * R;
    const uint32_t G = cus;
    w.rpx = ((S.npos + 7) / 8 + R - 1) / R * R;
    w.tiles_x = (w.rpx + RW - 1) / RW;
    w.G = G;
    const uint32_t spp = (w.tiles_x + G - 1) / G;
    const uint32_t slabs = ((ld >> 2) + group - 1) / group;
    w.nsweeps = slabs * spp;
    w.b_lo = b_lo; w.b_hi = b_hi;
    w.done = done;
    w.flags = flags | (accumulate ? 2u : 0u);
    w.split_partial = c.pieces;
    w.stat = ctl.stat;
    hipError_t e = hipMemsetAsync(done, 0, ((size_t)8 * w.nsweeps * (b_hi - b_lo) * 32 + 1) * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const dim3 gr(8u * slabs * spp * G), bl(SWEEP_NT);
#define GFS(GRP, HLV, RR, LD) hipLaunchKernelGGL((gatmh_forward_sweep_kernel<GRP, HLV, RR, LD>), gr, bl, 0, s, a, S, w, er, a_l, c.keys, c.dacc, c.den_slots, K, D, ldk)
#define GFS_R(GRP, HLV, LD)                                                                                             \
    do { if (R == 8) GFS(GRP, HLV, 8, LD); else if (R == 6) GFS(GRP, HLV, 6, LD); else if (R == 4) GFS(GRP, HLV, 4, LD); else GFS(GRP, HLV, 2, LD); } while (0)
#define GFS_R16(HLV) do { if (R == 4) GFS(16, HLV, 4, false); else GFS(16, HLV, 2, false); } while (0)
    if (group == 32) {
        if (HL == 2) GFS_R(32, 2, true); else if (HL == 4) GFS_R(32, 4, true); else if (HL == 8) GFS_R(32, 8, true); else GFS_R(32, 16, true);
    } else {
        if (R > 4) return hipErrorInvalidValue;
        if (HL == 2) GFS_R16(2); else if (HL == 4) GFS_R16(4); else if (HL == 8) GFS_R16(8); else GFS_R16(16);
    }
#undef GFS_R16
#undef GFS_R
#undef GFS
    return hipGetLastError();
}

// pieces of split rows, the self edge, the normalisation, m / den; then the rows whose denominator underflowed
hipError_t launch_gatmh_forward_sweep_finish(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const uint64_t *colptr,
                                             const uint32_t *rowidx, const BlockedAdj &S, const float *z, const float *zg, const float *el,
                                             const float *elg, const float *er, float *o, float *m, float *den, float *scratch,
                                             hipStream_t s) {
    if (N == 0) return hipSuccess;
    const GatSweepScratch c = gatmh_carve(scratch, S, N, ld, ldk);
    GatMhArgs a{N, K, D, ld, ldk, colptr, rowidx};
    if (S.nsplit)
        hipLaunchKernelGGL(gatmh_sweep_combine_kernel, dim3(S.nsplit), dim3(256), 0, s, S, ld, ldk, K, c.pieces, c.den_slots, o, c.dacc);
    const size_t n = (size_t)N * (ld >> 2);
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(gatmh_forward_finish_kernel, dim3(blocks), dim3(256), 0, s, a, z, el, er, c.keys, c.dacc, o, m, den, c.redo_flag, c.redo_list);
    hipLaunchKernelGGL(gatmh_forward_redo_kernel, dim3(64), dim3(256), 0, s, a, z, zg, el, elg, er, o, m, den, c.redo_flag, c.redo_list);
    return hipGetLastError();
}

}  // namespace dory
