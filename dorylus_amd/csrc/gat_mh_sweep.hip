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
// The second thing the linear form buys (round 5): the destination-side backward sweep disappears.  With alpha fixed,
//     t[v,k]   = sum_e alpha_e <dO[v,k,:], Z[src e,k,:]>              = <dO[v,k,:], O[v,k,:]>
//     der[v,k] = sum_e alpha_e (dalpha_e - t) l'_e,  l' in {1, 0.2}   = 0.8 (<dO[v,k,:], P[v,k,:]> - t dpos[v,k])
// with P = the part of O that came over edges on LeakyReLU's positive branch and dpos = their alpha mass -- two more sums
// the forward sweep carries along (tensors "op", "dpos"); t, der and the packed statistics then come from a row-wise
// kernel.  The source-side sweep uses the same split: del[u,k] = <Z[u,k,:], 0.2 S + 0.8 S+> - (0.2 T + 0.8 T+) with
// S = sum_v alpha dO_v (the main term of dZ), T = sum_v alpha t_v and their positive-branch parts -- no dot product, no
// cross-lane reduction per edge.
// Numerics of that split (review, round 5): del and der are DIFFERENCES of two sums of the size of t, not sums of alpha (dalpha - t).
// Where every dalpha of a row lies close to t (near-uniform attention, or attention on one edge) the true value is small
// against t and fp32 keeps it to eps |t|, not to eps |del|: |error(del, der)| <= ~32 eps max|t| on top of the usual 1e-4
// relative (tests/test_gpu_gat_mh.py::test_gat_mh_sweep_score_gradients_when_attention_is_flat_or_peaked checks exactly this bound
// in both regimes).  dz, dW and the sums da_l / da_r keep the relative criterion; the blocked kernels (gatmh_sweep = 0) sum edge by edge.
// LeakyReLU's kink: the branch of an (edge, head) is decided as t1 > t2 on the SHIFTED scores (magnitude |log2 den| ~ 5), i.e. to
// ~3e-7 absolute in el + er; an edge that close to zero may land on the other side than in the oracle, which moves del[u] / der[v]
// by 0.8 alpha (dalpha - t) and the forward value by < 3e-7.  Measured (round 6, a_l, a_r ~ 1e-3, 12 000 edges x 8 heads): a
// handful of such edges, 1.5e-2 relative on the del entries they touch -- a subgradient choice, not an accumulation error.
// Inside the sweep everything is in log2 units (a_l, er, m scaled by log2 e once): exp(s - m) = v_exp_f32(max(t1, t2)),
// t1 = el' + c1, t2 = 0.2 el' + c2 with c1 = er' - m', c2 = 0.2 er' - m' per (row, head) in an LDS table.
#include "gat_mh.hpp"
#include "sweep_core.hpp"

namespace dory {

#ifndef GATMH_SLACK
#define GATMH_SLACK SWEEP_SLACK   // windows a workgroup may run ahead of its sweep's slowest: 0 / 1 / 2 = 20.5 / 18.0 / 18.4 ms per 8-head epoch (round 5)
#endif
#ifndef GATMH_SRC16_LOADER
#define GATMH_SRC16_LOADER true
#endif
#ifndef GATMH_SRC16_BATCH
#define GATMH_SRC16_BATCH 4   // (round 6, with one statistics gather per batch: 2 -> 4 = 2.91 -> 2.77 ms; 2 was best while every entry had its own)
#endif
#ifndef GATMH_FWD16_ROWS
#define GATMH_FWD16_ROWS 2
#endif
#ifndef GATMH_FWD_BATCH
#define GATMH_FWD_BATCH SWEEP_U
#endif
#ifndef GATMH_FWD16_BATCH
#define GATMH_FWD16_BATCH SWEEP_U
#endif
// forward: the source's score el[u,k] -- 0: formed from the gathered row (<z_u, a_l>: two packed multiplies, an add and log2(HL)
// DPP adds per entry); 1: fetched from the el table, one 4-byte gather per batch of entries (lane j of a quad fetches entry j's,
// DPP quad broadcast) -- the same batching as GATMH_SRC_AUX_MODE 3
// Measured (round 6): 128-float launch 4.04 -> 4.91 ms with the table (the extra gather and its LDS index read cost more than
// the five vector instructions they replace: the forward is not bound by its arithmetic), 64-float launch 2.51 -> 2.46 ms.
// Value = the widest lane group that uses the table: 16 = the 64-float launches only.
#ifndef GATMH_FWD_EL_TABLE
#define GATMH_FWD_EL_TABLE 16
#endif
#ifndef GATMH_SRC16_ROWS
#define GATMH_SRC16_ROWS 2
#endif
#ifndef GATMH_SRC_BATCH
#define GATMH_SRC_BATCH 3
#endif
// The source side's second gather (the destination's (c1', c2', t) record; the HL lanes of a head want the same 12 bytes):
//   0  every lane loads it, once per entry (round 5)
//   3  ONE instruction per BATCH of entries: the addresser's cost is per instruction (tools/probes/aux_gather_probe.hip), so
//      lane j of a quad fetches the record of the batch's entry j and the quad's lanes get entry u's record by a DPP quad
//      broadcast (heads of two lanes: two entries per instruction)
// (1 / 2 -- only the quad leaders load, under the EXEC mask or with the other lanes out of range -- were measured and removed:
// 5.25 / 5.09 against 4.96 ms.)  Measured (round 6, Reddit-large, 8 heads; profiles/r06_gatmh_aux_forms.txt): 32-lane launch
// 4.96 (0) / 4.87 ms (3); 16-lane launch with batches of four 2.91 -> 2.77 ms (3).  The probe's 30 % (one gather instruction in
// four gone) does not arrive in the sweep: its steps are bound by the chain LDS -> gathers -> sums of 16 waves, not by the addresser.
#ifndef GATMH_SRC_AUX_MODE
#define GATMH_SRC_AUX_MODE 3
#endif
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

// ---- forward: acc[v,:] = sum_e exp(s_e - m_v) z[src(e),:],  den[v,k] = sum_e exp(s_e - m_v), and the positive-branch parts
// of both (self edge, normalisation: finish kernel) ----------------------------------------------------------------------
template <int GROUP, int HL, int R>
struct GatFwdSweepOp {
    static constexpr bool PLAIN = false, UNIT_W = true, PROLOGUE = true, AUX_BATCH = (GROUP <= GATMH_FWD_EL_TABLE);
    static constexpr int EPL = HL >= 4 ? 4 : 2;
    static constexpr int BATCH = GROUP == 16 ? GATMH_FWD16_BATCH : GATMH_FWD_BATCH;
    static constexpr int SLACK = GATMH_SLACK;
    static constexpr int HPS = GROUP / HL;                     // heads per slab of GROUP lanes
    static constexpr int RW = (SWEEP_NT / GROUP) * R;
    // arguments
    const float *er, *a_l;
    const int *elmax_key;
    float *accp;        // [N][ld]: unnormalised positive-branch sums (the finish kernel turns them into "op")
    float *dacc;        // [N][2 ldk]: unnormalised (den, dpos) pairs (the finish kernel adds the self edge, writes den / dpos)
    float *pos_slots, *den_slots;   // the same for pieces of split rows: [nslots][ld], [nslots][2 ldk]
    uint32_t K, D, ldk;
    const float *el, *elg;          // GATMH_FWD_EL_TABLE: the sources' scores (local rows, ghost rows)
    // per thread
    float4 al4;
    uint32_t k, hl, aux_b, qpos;
    __amdgpu_buffer_rsrc_t rs2;
    const float2 *ctab;
    typedef float f2 __attribute__((ext_vector_type(2)));
    struct Row { float4 acc, accp; f2 den; };                  // den = (all edges, positive-branch edges)
    struct RowC { float c1, c2; };
    typedef float Aux;                                          // AUX_BATCH: el[src, k]
    __device__ __forceinline__ Aux aux(uint32_t, uint32_t, bool) const { return 0.f; }
    template <int U0, int NB, int NL>
    __device__ __forceinline__ void aux_spread(const float (&rec)[NL], Aux (&ax)[NB]) const {
        if constexpr (U0 < NB) {
            constexpr int pp = U0 % EPL;
            constexpr int QP = HL >= 4 ? pp * 0x55 : (pp | (pp << 2) | ((2 + pp) << 4) | ((2 + pp) << 6));
            ax[U0] = sw_dpp<QP>(rec[U0 / EPL]);
            aux_spread<U0 + 1, NB, NL>(rec, ax);
        }
    }
    template <int NB> __device__ __forceinline__ void aux_batch(const uint2 *stp, uint32_t n, Aux (&ax)[NB]) const {
        constexpr int NL = (NB + EPL - 1) / EPL;
        float rec[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const uint32_t j = (uint32_t)i * EPL + qpos;
            const uint32_t sidx = stp[j < (uint32_t)NB ? j : (uint32_t)NB - 1].x;
            rec[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs2, j < n ? __umul24(sidx, ldk * 4u) + aux_b : 0xFFFFFFFFu, 0, 0));
        }
        aux_spread<0, NB, NL>(rec, ax);
    }
    __device__ __forceinline__ void init(Row &r) const {
        r.acc = make_float4(0.f, 0.f, 0.f, 0.f); r.accp = r.acc; r.den = (f2){0.f, 0.f};
    }
    __device__ __forceinline__ void prologue(const SpmmArgs &a, const BlockedAdj &B, uint32_t pos0, uint32_t xend, bool ghost_launch, uint32_t col, int li) {
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
        if constexpr (AUX_BATCH) {
            rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(ghost_launch ? elg : el), 0, (ghost_launch ? B.nghost : a.N) * ldk * 4u, 0x00020000);
            aux_b = k * 4u - (ghost_launch ? a.N : 0u) * ldk * 4u;
            qpos = (uint32_t)li & (HL >= 4 ? 3u : 1u);
        }
        const uint32_t f0 = col * 4, KD = K * D;               // a_l is a dense K x D vector (41-feature heads end mid-float4)
        al4 = make_float4(f0 < KD ? a_l[f0] * GATMH_LOG2E : 0.f, f0 + 1 < KD ? a_l[f0 + 1] * GATMH_LOG2E : 0.f,
                          f0 + 2 < KD ? a_l[f0 + 2] * GATMH_LOG2E : 0.f, f0 + 3 < KD ? a_l[f0 + 3] * GATMH_LOG2E : 0.f);
    }
    __device__ __forceinline__ RowC row_const(uint32_t lrow) const {
        const float2 c = ctab[lrow * HPS + hl];
        return RowC{c.x, c.y};
    }
    // The sweep is bound by the vector ALU as much as by the addresser (16 vector instructions per gather instruction in the
    // first cut: 3.78 ms per 128-float launch whatever the rows per group or the gates; 13 with the packed forms: 3.39 ms):
    // everything here is written for the packed fp32 instructions (v_pk_mul / v_pk_fma: two lanes' worth per issue slot).
    template <bool FULL>
    __device__ __forceinline__ void entry(Row &r, const RowC &c, const float4 &x, Aux ax, bool on) const {
        float e;
        f2 kk;
        if constexpr (AUX_BATCH) {
            e = ax;                                                                   // el[src] as the scores kernel wrote it
            kk = (f2){GATMH_LOG2E, GATMH_SLOPE * GATMH_LOG2E};
        } else {
            const f2 xlo = {x.x, x.y}, xhi = {x.z, x.w}, alo = {al4.x, al4.y}, ahi = {al4.z, al4.w};
            const f2 p = __builtin_elementwise_fma(xhi, ahi, xlo * alo);             // v_pk_mul + v_pk_fma
            e = sw_head_sum<HL>(p.x + p.y);                                           // el'[src] of this lane's head
            kk = (f2){1.f, GATMH_SLOPE};
        }
        const f2 ee = {e, e}, cc = {c.c1, c.c2};
        const f2 t = __builtin_elementwise_fma(ee, kk, cc);                            // (e + c1, 0.2 e + c2): one v_pk_fma
        float al = __builtin_amdgcn_exp2f(fmaxf(t.x, t.y));
        if constexpr (!FULL) al = on ? al : 0.f;               // (an absent slot gathered zeros: its score is not zero)
        const float alp = t.x > t.y ? al : 0.f;                // positive branch: el + er > 0  <=>  t1 > t2
        r.den += (f2){al, alp};
        r.acc = fma4(al, x, r.acc);
        r.accp = fma4(alp, x, r.accp);
    }
    __device__ __forceinline__ void store(const Row &r, const SpmmArgs &a, const SweepArgs &w, uint32_t v, bool piece, uint32_t slot,
                                          uint32_t col, uint32_t nchunk, const float4 *) const {
        float4 *q = piece ? reinterpret_cast<float4 *>(w.split_partial) + (size_t)slot * nchunk + col
                          : reinterpret_cast<float4 *>(a.out) + (size_t)v * nchunk + col;
        float4 *qp = piece ? reinterpret_cast<float4 *>(pos_slots) + (size_t)slot * nchunk + col
                           : reinterpret_cast<float4 *>(accp) + (size_t)v * nchunk + col;
        float2 *dq = reinterpret_cast<float2 *>(piece ? den_slots + (size_t)slot * 2 * ldk : dacc + (size_t)v * 2 * ldk) + k;
        float4 o4 = r.acc, p4 = r.accp;
        float2 dn = make_float2(r.den.x, r.den.y);
        const bool head_lane = (threadIdx.x % HL) == 0 && col * 4 < K * D;
        if (piece ? (w.flags & 2u) != 0 : a.accumulate != 0) {   // second launch of a partitioned run (ghost blocks)
            const float4 p = *q, pp = *qp;
            o4.x += p.x; o4.y += p.y; o4.z += p.z; o4.w += p.w;
            p4.x += pp.x; p4.y += pp.y; p4.z += pp.z; p4.w += pp.w;
            if (head_lane) { const float2 d0 = *dq; dn.x += d0.x; dn.y += d0.y; }
        }
        *q = o4;
        *qp = p4;
        if (head_lane) *dq = dn;
    }
};

template <int GROUP, int HL, int R, bool LOADER>
__global__ __launch_bounds__(SWEEP_NT) void gatmh_forward_sweep_kernel(SpmmArgs a, BlockedAdj B, SweepArgs w, const float *er,
                                                                       const float *a_l, const int *elmax_key, float *accp, float *dacc,
                                                                       float *pos_slots, float *den_slots, uint32_t K, uint32_t D,
                                                                       uint32_t ldk, const float *el, const float *elg) {
    GatFwdSweepOp<GROUP, HL, R> op{er, a_l, elmax_key, accp, dacc, pos_slots, den_slots, K, D, ldk, el, elg};
    sweep_run<GROUP, R, false, LOADER>(a, B, w, op);
}

// pieces of split rows: dst[v,:] = sum of the pieces' slots (piece order), for a row tensor (width ld) and a per-head one
__global__ __launch_bounds__(256) void gatmh_sweep_combine_kernel(BlockedAdj B, uint32_t ld, uint32_t ldh, const float *part,
                                                                  const float *head_slots, float *rows_out, float *heads_out) {
    const uint32_t sr = blockIdx.x;
    if (sr >= B.nsplit) return;
    const uint32_t v = B.split_rows[3 * sr], s0 = B.split_rows[3 * sr + 1], P = B.split_rows[3 * sr + 2];
    for (uint32_t f = threadIdx.x; f < ld + ldh; f += blockDim.x) {
        float s = 0.f;
        if (f < ld) {
            for (uint32_t p = 0; p < P; ++p) s += part[(size_t)(s0 + p) * ld + f];
            rows_out[(size_t)v * ld + f] = s;
        } else {
            for (uint32_t p = 0; p < P; ++p) s += head_slots[(size_t)(s0 + p) * ldh + (f - ld)];
            heads_out[(size_t)v * ldh + (f - ld)] = s;
        }
    }
}

// o = (acc + e_self z_v) / (dacc + e_self), op / dpos likewise with the self edge on its branch; m, den for the backward
// passes; rows whose denominator underflowed are listed
__global__ __launch_bounds__(256) void gatmh_forward_finish_kernel(GatMhArgs a, const float *z, const float *el, const float *er,
                                                                   const int *elmax_key, const float *dacc, float *o, float *op,
                                                                   float *m_out, float *den_out, float *dpos_out, uint32_t *redo_flag,
                                                                   uint32_t *redo_list /*[0] = count*/) {
    const uint32_t nchunk = a.ld >> 2;
    const size_t n = (size_t)a.N * nchunk;
    const float4 *z4 = reinterpret_cast<const float4 *>(z);
    float4 *o4 = reinterpret_cast<float4 *>(o), *p4 = reinterpret_cast<float4 *>(op);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t v = (uint32_t)(i / nchunk), col = (uint32_t)(i % nchunk);
        if (col * 4 >= a.K * a.D) continue;
        const uint32_t k = min((col * 4) / a.D, a.K - 1);
        const size_t vk = (size_t)v * a.ldk + k;
        const float e = er[vk], pre = el[vk] + e;
        const float mm = lrelu02(gm_fkey_inv(elmax_key[k]) + e);
        const float es = __builtin_amdgcn_exp2f((lrelu02(pre) - mm) * GATMH_LOG2E);
        const float esp = pre > 0.f ? es : 0.f;
        const float2 d2 = reinterpret_cast<const float2 *>(dacc + (size_t)v * 2 * a.ldk)[k];
        const float dn = d2.x + es;
        const float idn = 1.f / dn;
        const float4 acc = o4[i], accp = p4[i], x = z4[i];
        o4[i] = make_float4(fmaf(es, x.x, acc.x) * idn, fmaf(es, x.y, acc.y) * idn, fmaf(es, x.z, acc.z) * idn, fmaf(es, x.w, acc.w) * idn);
        p4[i] = make_float4(fmaf(esp, x.x, accp.x) * idn, fmaf(esp, x.y, accp.y) * idn, fmaf(esp, x.z, accp.z) * idn, fmaf(esp, x.w, accp.w) * idn);
        if ((col * 4) % a.D < 4 || a.K == 1) {
            if (a.K != 1 || col == 0) { m_out[vk] = mm; den_out[vk] = dn; dpos_out[vk] = (d2.y + esp) * idn; }
            if (!(dn >= GATMH_DEN_TINY) && atomicExch(redo_flag + v, 1u) == 0u) redo_list[1 + atomicAdd(redo_list, 1u)] = v;
        }
    }
}

// the rows the finish kernel listed, recomputed with their own maximum (online softmax over the row's in-edges and the
// self edge, one wave per row, lanes over the features); rare by construction, so nothing here is tuned
__global__ __launch_bounds__(256) void gatmh_forward_redo_kernel(GatMhArgs a, const float *z, const float *zg, const float *el,
                                                                 const float *elg, const float *er, float *o, float *op, float *m_out,
                                                                 float *den_out, float *dpos_out, uint32_t *redo_flag,
                                                                 const uint32_t *redo_list) {
    const uint32_t cnt = redo_list[0];
    const int lane = threadIdx.x & 63;
    const uint32_t KD = a.K * a.D;
    for (uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6); i < cnt; i += gridDim.x * 4) {
        const uint32_t v = redo_list[1 + i];
        for (uint32_t f = (uint32_t)lane; f < KD; f += 64) {
            const uint32_t k = f / a.D;
            const float er_v = er[(size_t)v * a.ldk + k];
            float mx = -INFINITY, den = 0.f, denp = 0.f, acc = 0.f, accp = 0.f;
            const uint64_t e0 = a.ptr[v], e1 = a.ptr[v + 1];
            for (uint64_t e = e0; e <= e1; ++e) {                 // e1 stands for the self edge
                const uint32_t u = e < e1 ? a.idx[e] : v;
                const bool loc = u < a.N;
                const float el_u = loc ? el[(size_t)u * a.ldk + k] : elg[(size_t)(u - a.N) * a.ldk + k];
                const float zu = loc ? z[(size_t)u * a.ld + f] : zg[(size_t)(u - a.N) * a.ld + f];
                const float pre = el_u + er_v, s = lrelu02(pre);
                const float mn = fmaxf(mx, s);
                const float sc = __expf(mx - mn), al = __expf(s - mn);   // (exp(-inf) = 0 on the first edge)
                const float alp = pre > 0.f ? al : 0.f;
                acc = fmaf(acc, sc, al * zu);
                accp = fmaf(accp, sc, alp * zu);
                den = fmaf(den, sc, al);
                denp = fmaf(denp, sc, alp);
                mx = mn;
            }
            o[(size_t)v * a.ld + f] = acc / den;
            op[(size_t)v * a.ld + f] = accp / den;
            if (f % a.D == 0) {
                m_out[(size_t)v * a.ldk + k] = mx; den_out[(size_t)v * a.ldk + k] = den; dpos_out[(size_t)v * a.ldk + k] = denp / den;
            }
        }
        if (lane == 0) redo_flag[v] = 0u;
    }
}

// ---- backward, destination side: no edges (header).  t = <dO, O>, der = 0.8 (<dO, P> - t dpos), st4 = (er, m, 1/den, t):
// one float4 of a row per thread, the head's HL threads reduced by shuffles (HL divides the float4s per row).
__global__ __launch_bounds__(256) void gatmh_dst_rowwise_kernel(GatMhArgs a, int HL, const float *d_o, const float *o, const float *op,
                                                                const float *dpos, const float *er, const float *m, const float *den,
                                                                float *t_out, float *der_out, float4 *st4, uint32_t lds4) {
    const uint32_t nchunk = a.ld >> 2;
    const size_t n = (size_t)a.N * nchunk;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i < n;
    const size_t ii = ok ? i : 0;
    const uint32_t v = (uint32_t)(ii / nchunk), col = (uint32_t)(ii % nchunk);
    const bool live = ok && col * 4 < a.K * a.D;
    const float4 g = reinterpret_cast<const float4 *>(d_o)[ii], x = reinterpret_cast<const float4 *>(o)[ii],
                 p = reinterpret_cast<const float4 *>(op)[ii];
    float tt = live ? fmaf(g.x, x.x, fmaf(g.y, x.y, fmaf(g.z, x.z, g.w * x.w))) : 0.f;     // (columns past K*D are zero in all three)
    float tp = live ? fmaf(g.x, p.x, fmaf(g.y, p.y, fmaf(g.z, p.z, g.w * p.w))) : 0.f;
    for (int off = 1; off < HL; off <<= 1) { tt += __shfl_xor(tt, off, 64); tp += __shfl_xor(tp, off, 64); }
    if (live && (col % (uint32_t)HL) == 0) {
        const uint32_t k = min((col * 4) / a.D, a.K - 1);
        const size_t vk = (size_t)v * a.ldk + k;
        t_out[vk] = tt;
        der_out[vk] = (1.f - GATMH_SLOPE) * (tp - tt * dpos[vk]);
        st4[(size_t)v * lds4 + k] = make_float4(er[vk], m[vk], 1.f / den[vk], tt);
    }
}

// ---- backward, source side, on the skeleton: rows = sources u (out-edges), entries = destinations v ---------------------------
// stx[v,k] = (c1', c2', t): the destination's side of the score in log2 units with 1/den folded in,
//     alpha_uv = exp2(max(el'_u + c1'_v, 0.2 el'_u + c2'_v)),   c1' = (er - m) log2e + log2(1/den),  c2' = (0.2 er - m) log2e + log2(1/den)
__global__ void gatmh_stx_kernel(uint64_t n /*rows x K*/, uint32_t K, const float4 *st4, uint32_t lds4, float4 *stx) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 s = st4[(i / K) * lds4 + i % K];       // (er, m, 1/den, t)
    const float li = __log2f(s.z);
    stx[i] = make_float4(fmaf(s.x - s.y, GATMH_LOG2E, li), fmaf(GATMH_SLOPE * s.x - s.y, GATMH_LOG2E, li), s.w, 0.f);
}

template <int GROUP, int HL, int R>
struct GatSrcSweepOp {
    // (the destinations' statistics fetched once per batch through the LDS crossbar instead of once per entry: measured, no gain --
    // profiles/r05_gatmh_src_aux_batch_experiment.patch)
    static constexpr bool PLAIN = false, UNIT_W = true, PROLOGUE = true, AUX_BATCH = GATMH_SRC_AUX_MODE == 3;
    static constexpr int BATCH = GROUP == 16 ? GATMH_SRC16_BATCH : GATMH_SRC_BATCH;   // two gathers per entry (rows, statistics); 16-lane groups: four lane groups per gather instruction
    static constexpr int EPL = HL >= 4 ? 4 : 2;    // AUX_BATCH: entries one statistics gather serves (the lanes of a quad that share a head)
    static constexpr int SLACK = GATMH_SLACK;
    static constexpr int HPS = GROUP / HL;
    static constexpr int RW = (SWEEP_NT / GROUP) * R;
    // arguments
    const float *el;                // [N][ldk] of the sources (= the rows)
    const float4 *stx, *stxg;       // [N][K], [Gdst][K]
    float *sp;                      // [N][ld]: positive-branch part of S
    float *tacc;                    // [N][2 ldk]: (T, T+)
    float *pos_slots, *t_slots;     // pieces of split rows
    uint32_t K, D, ldk, N, G;
    // per thread
    uint32_t k, hl, aux_b, qpos;
    __amdgpu_buffer_rsrc_t rs2;
    const float2 *etab;
    typedef float f2 __attribute__((ext_vector_type(2)));
    struct Row { float4 s, sp; f2 tt; };
    struct RowC { f2 e; };          // (el'_u, 0.2 el'_u)
    typedef float3 Aux;             // (c1', c2', t): a 12-byte load (a register less per gather in flight than the 16 bytes)
    __device__ __forceinline__ void init(Row &r) const {
        r.s = make_float4(0.f, 0.f, 0.f, 0.f); r.sp = r.s; r.tt = (f2){0.f, 0.f};
    }
    __device__ __forceinline__ void prologue(const SpmmArgs &a, const BlockedAdj &B, uint32_t pos0, uint32_t xend, bool ghost_launch, uint32_t col, int li) {
        __shared__ float2 tab[RW * HPS];
        const uint32_t head0 = (col / GROUP) * HPS;
        for (uint32_t i = threadIdx.x; i < (uint32_t)(RW * HPS); i += SWEEP_NT) {
            const uint32_t lrow = i / HPS, kk = head0 + i % HPS, pos = pos0 + lrow;
            const uint32_t u = pos < xend ? (B.perm ? B.perm[pos] : pos) : 0xFFFFFFFFu;
            float2 c = make_float2(0.f, 0.f);
            if (u != 0xFFFFFFFFu && kk < K) {
                const float e = el[(size_t)u * ldk + kk] * GATMH_LOG2E;
                c = make_float2(e, GATMH_SLOPE * e);
            }
            tab[i] = c;
        }
        etab = tab;
        hl = (uint32_t)li / HL;
        k = min(head0 + hl, K - 1);
        // the destinations' packed statistics through a buffer resource of their own: 16 bytes per (v, k)
        const uint32_t rowb = K * 16u;
        rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(ghost_launch ? stxg : stx), 0, (ghost_launch ? G : N) * rowb, 0x00020000);
        aux_b = k * 16u - (ghost_launch ? N : 0u) * rowb;
        qpos = (uint32_t)li & (HL >= 4 ? 3u : 1u);
    }
    __device__ __forceinline__ RowC row_const(uint32_t lrow) const {
        const float2 c = etab[lrow * HPS + hl];
        return RowC{(f2){c.x, c.y}};
    }
    __device__ __forceinline__ Aux aux(uint32_t sidx, uint32_t, bool on) const {
        typedef uint32_t u3 __attribute__((ext_vector_type(3)));
        const u3 v = __builtin_amdgcn_raw_buffer_load_b96(rs2, on ? __umul24(sidx, K * 16u) + aux_b : 0xFFFFFFFFu, 0, 0);
        return make_float3(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z));
    }
    // AUX_BATCH: the records of the batch's entries [0, n) -- lane (quad position j) fetches entry i * EPL + j with load i
    template <int U0, int NB, int NL>
    __device__ __forceinline__ void aux_spread(const float3 (&rec)[NL], Aux (&ax)[NB]) const {
        if constexpr (U0 < NB) {
            constexpr int pp = U0 % EPL;
            constexpr int QP = HL >= 4 ? pp * 0x55 : (pp | (pp << 2) | ((2 + pp) << 4) | ((2 + pp) << 6));
            const float3 &v = rec[U0 / EPL];
            ax[U0] = make_float3(sw_dpp<QP>(v.x), sw_dpp<QP>(v.y), sw_dpp<QP>(v.z));
            aux_spread<U0 + 1, NB, NL>(rec, ax);
        }
    }
    template <int NB> __device__ __forceinline__ void aux_batch(const uint2 *stp, uint32_t n, Aux (&ax)[NB]) const {
        typedef uint32_t u3 __attribute__((ext_vector_type(3)));
        constexpr int NL = (NB + EPL - 1) / EPL;
        float3 rec[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const uint32_t j = (uint32_t)i * EPL + qpos;
            const uint32_t sidx = stp[j < (uint32_t)NB ? j : (uint32_t)NB - 1].x;       // (one LDS read per lane: four addresses per head)
            const u3 v = __builtin_amdgcn_raw_buffer_load_b96(rs2, j < n ? __umul24(sidx, K * 16u) + aux_b : 0xFFFFFFFFu, 0, 0);
            rec[i] = make_float3(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z));
        }
        aux_spread<0, NB, NL>(rec, ax);
    }
    template <bool FULL>
    __device__ __forceinline__ void entry(Row &r, const RowC &c, const float4 &x, const Aux &sv, bool on) const {
        const f2 t = c.e + (f2){sv.x, sv.y};                     // (el' + c1', 0.2 el' + c2')
        float al = __builtin_amdgcn_exp2f(fmaxf(t.x, t.y));
        if constexpr (!FULL) al = on ? al : 0.f;                 // (an absent slot read zeros: exp2(el') is not zero)
        const float alp = t.x > t.y ? al : 0.f;
        r.tt = __builtin_elementwise_fma((f2){al, alp}, (f2){sv.z, sv.z}, r.tt);
        r.s = fma4(al, x, r.s);
        r.sp = fma4(alp, x, r.sp);
    }
    __device__ __forceinline__ void store(const Row &r, const SpmmArgs &a, const SweepArgs &w, uint32_t u, bool piece, uint32_t slot,
                                          uint32_t col, uint32_t nchunk, const float4 *) const {
        float4 *q = piece ? reinterpret_cast<float4 *>(w.split_partial) + (size_t)slot * nchunk + col
                          : reinterpret_cast<float4 *>(a.out) + (size_t)u * nchunk + col;
        float4 *qp = piece ? reinterpret_cast<float4 *>(pos_slots) + (size_t)slot * nchunk + col
                           : reinterpret_cast<float4 *>(sp) + (size_t)u * nchunk + col;
        float2 *tq = reinterpret_cast<float2 *>(piece ? t_slots + (size_t)slot * 2 * ldk : tacc + (size_t)u * 2 * ldk) + k;
        float4 s4 = r.s, p4 = r.sp;
        float2 t2 = make_float2(r.tt.x, r.tt.y);
        const bool head_lane = (threadIdx.x % HL) == 0 && col * 4 < K * D;
        if (piece ? (w.flags & 2u) != 0 : a.accumulate != 0) {
            const float4 p = *q, pp = *qp;
            s4.x += p.x; s4.y += p.y; s4.z += p.z; s4.w += p.w;
            p4.x += pp.x; p4.y += pp.y; p4.z += pp.z; p4.w += pp.w;
            if (head_lane) { const float2 d0 = *tq; t2.x += d0.x; t2.y += d0.y; }
        }
        *q = s4;
        *qp = p4;
        if (head_lane) *tq = t2;
    }
};

template <int GROUP, int HL, int R, bool LOADER>
__global__ __launch_bounds__(SWEEP_NT) void gatmh_src_sweep_kernel(SpmmArgs a, BlockedAdj B, SweepArgs w, const float *el,
                                                                   const float4 *stx, const float4 *stxg, float *sp, float *tacc,
                                                                   float *pos_slots, float *t_slots, uint32_t K, uint32_t D, uint32_t ldk,
                                                                   uint32_t G) {
    GatSrcSweepOp<GROUP, HL, R> op{el, stx, stxg, sp, tacc, pos_slots, t_slots, K, D, ldk, a.N, G};
    sweep_run<GROUP, R, false, LOADER>(a, B, w, op);
}

// dz[u,:] = S + alpha_self dO[u,:] + del[u,k] a_l + der[u,k] a_r,   del[u,k] = <Z[u,k,:], 0.2 S' + 0.8 S+'> - (0.2 T' + 0.8 T+')
// (primes: with the self edge); one float4 of a row per thread, the head's HL threads reduced by shuffles
__global__ __launch_bounds__(256) void gatmh_src_finish_kernel(GatMhArgs a, int HL, const float *z, const float *el, const float4 *stx,
                                                               const float *d_o, const float *sp, const float *tacc, const float *der,
                                                               const float *a_l, const float *a_r, float *dz, float *del_out) {
    const uint32_t nchunk = a.ld >> 2;
    const size_t n = (size_t)a.N * nchunk;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i < n;
    const size_t ii = ok ? i : 0;
    const uint32_t u = (uint32_t)(ii / nchunk), col = (uint32_t)(ii % nchunk);
    const bool live = ok && col * 4 < a.K * a.D;
    const uint32_t k = min((col * 4) / a.D, a.K - 1);
    const size_t uk = (size_t)u * a.ldk + k;
    const float4 sv = stx[(size_t)u * a.K + k];
    const float e = el[uk] * GATMH_LOG2E;
    const float t1 = e + sv.x, t2 = fmaf(e, GATMH_SLOPE, sv.y);
    const float as = __builtin_amdgcn_exp2f(fmaxf(t1, t2)), asp = t1 > t2 ? as : 0.f;       // the self edge
    const float4 g = reinterpret_cast<const float4 *>(d_o)[ii], zz = reinterpret_cast<const float4 *>(z)[ii];
    float4 s = reinterpret_cast<const float4 *>(dz)[ii], p = reinterpret_cast<const float4 *>(sp)[ii];
    s = make_float4(fmaf(as, g.x, s.x), fmaf(as, g.y, s.y), fmaf(as, g.z, s.z), fmaf(as, g.w, s.w));
    p = make_float4(fmaf(asp, g.x, p.x), fmaf(asp, g.y, p.y), fmaf(asp, g.z, p.z), fmaf(asp, g.w, p.w));
    const float lo = GATMH_SLOPE, hi = 1.f - GATMH_SLOPE;
    float dd = live ? zz.x * fmaf(hi, p.x, lo * s.x) + zz.y * fmaf(hi, p.y, lo * s.y) + zz.z * fmaf(hi, p.z, lo * s.z) + zz.w * fmaf(hi, p.w, lo * s.w) : 0.f;
    for (int off = 1; off < HL; off <<= 1) dd += __shfl_xor(dd, off, 64);
    const float2 tt = reinterpret_cast<const float2 *>(tacc + (size_t)u * 2 * a.ldk)[k];
    const float del = dd - (lo * fmaf(as, sv.z, tt.x) + hi * fmaf(asp, sv.z, tt.y));
    if (!live) return;
    const float dr = der[uk];
    float r[4] = {s.x, s.y, s.z, s.w};
    float *out = dz + (size_t)u * a.ld + (size_t)col * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t f = col * 4 + c;
        out[c] = f < a.K * a.D ? r[c] + del * a_l[f] + dr * a_r[f] : 0.f;
    }
    if ((col % (uint32_t)HL) == 0) del_out[uk] = del;
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

// rows per lane group of a launch.  The layout is dealt for the 32-lane launches; a launch walks its positions with as many
// rows per group as its kernel holds in registers WITHOUT SPILLING: a sweep kernel sits at one 1024-thread workgroup per
// CU (128 registers per lane), its inner chain is LDS -> gathers -> sums, and a scratch access in that chain is a dependent
// miss per step (K1s's loader kernels lost 2 % to six spilled addresses, round 3).  The table below is what hipcc 7.2
// allocates; tests/test_kernel_resources.py reads the code objects of the built library and fails if a variant these
// rules can select spills.  (Round 5 first blamed spilling variants for intermittent wrong sums with four contexts on
// one device.  The cause was in the test harness: torch's zero fill of the transport buffers, on the NULL stream, racing
// the pack kernel on the context's non-blocking stream -- slower kernels only made the window wider.  Fixed there.)
// pass: 0 forward, 1 source side.
int gatmh_sweep_rows(const BlockedAdj &S, int group, int HL, int pass) {
    const int r = (int)S.rows_per_group;
    // 16-lane launches (64-float layers) have twice the lane groups per workgroup: half the layout's rows per group walks the
    // same rows per workgroup and step as the layout was dealt for.  Measured (round 5, Reddit-large, 8 heads, loader wave on):
    // forward 2 / 4 rows = 2.50 / 2.77 ms, source side 2 / 4 / 6 rows = 2.91 / 3.01 / 3.50 ms
    if (group == 16) return std::max(2, std::min(r / 2, pass == 0 ? GATMH_FWD16_ROWS : GATMH_SRC16_ROWS));
    // ten accumulator registers per row (the per-(row, head) constants already sit in the LDS table): 4 rows are what fits 128
    // registers with batches of 3-4.  Round 6 (profiles/r06_gatmh_aux_forms.txt): 6 rows fit without spills only with batches of
    // two (five sweeps per XCD instead of eight) and run the 128-float forward at 4.77 instead of 4.09 ms; 8 rows spill 18-32
    // registers at any batch size
    const int cap = pass == 0 ? 4 : (HL != 16 ? 4 : 2);
    int R = std::min(r, cap);
    if (R == 3) R = 2;
    return R;
}

// scratch layout of one layer's forward (floats): [pieces: nslots x ld][positive-branch pieces: nslots x ld][den slots: nslots x 2 ldk]
// [dacc: N x 2 ldk][keys: 64][redo flags: N][redo list: 1 + N]
size_t gatmh_sweep_scratch_bytes(const BlockedAdj &S, uint32_t N, uint32_t ld, uint32_t ldk) {
    return ((size_t)S.nslots * (2 * ld + 2 * ldk) + (size_t)N * 2 * ldk + 64 + (size_t)N + 1 + (size_t)N) * sizeof(float) + 256;
}
struct GatSweepScratch {
    float *pieces, *pos_slots, *den_slots, *dacc;
    int *keys;
    uint32_t *redo_flag, *redo_list;
};
static GatSweepScratch gatmh_carve(float *scratch, const BlockedAdj &S, uint32_t N, uint32_t ld, uint32_t ldk) {
    GatSweepScratch c;
    c.pieces = scratch;
    c.pos_slots = c.pieces + (size_t)S.nslots * ld;
    c.den_slots = c.pos_slots + (size_t)S.nslots * ld;
    c.dacc = c.den_slots + (size_t)S.nslots * 2 * ldk;
    c.keys = reinterpret_cast<int *>(c.dacc + (size_t)N * 2 * ldk);
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

// geometry of one launch over the source blocks [b_lo, b_hi) of a sweep layout (as launch_spmm_sweep)
static bool gatmh_sweep_geom(const SpmmArgs &a, const BlockedAdj &S, int group, int R, uint32_t cus, uint32_t b_lo, uint32_t b_hi,
                             bool accumulate, uint32_t *done, const SweepCtl &ctl, uint32_t flags, float *pieces, SweepArgs *w, dim3 *grid) {
    if (!sweep_supported(a, S, group) || b_hi > S.nb || cus == 0 || cus > 32 || !ctl.stat) return false;
    if (b_lo < S.nb_local && b_hi > S.nb_local) return false;
    if (b_lo >= S.nb_local && !a.xg) return false;
    if (R != 8 && R != 6 && R != 4 && R != 2) return false;
    const uint32_t RW = (uint32_t)(SWEEP_NT / group) * R;
    *w = SweepArgs{};
    w->rpx = ((S.npos + 7) / 8 + R - 1) / R * R;
    w->tiles_x = (w->rpx + RW - 1) / RW;
    w->G = cus;
    const uint32_t spp = (w->tiles_x + cus - 1) / cus;
    const uint32_t slabs = ((a.ld >> 2) + group - 1) / group;
    w->nsweeps = slabs * spp;
    w->b_lo = b_lo; w->b_hi = b_hi;
    w->done = done;
    w->flags = flags | (accumulate ? 2u : 0u);
    w->split_partial = pieces;
    w->stat = ctl.stat;
    *grid = dim3(8u * slabs * spp * cus);
    return true;
}

// one launch over the source blocks [b_lo, b_hi) of the sweep layout S (a partition with ghost rows: local-source blocks
// first, then the ghost blocks with accumulate = true)
hipError_t launch_gatmh_forward_sweep_part(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const BlockedAdj &S,
                                           const float *z, const float *zg, const float *er, const float *a_l, float *o, float *op,
                                           float *scratch, uint32_t cus, uint32_t b_lo, uint32_t b_hi, bool accumulate, uint32_t *done,
                                           const SweepCtl &ctl, uint32_t flags, hipStream_t s, const float *el, const float *elg) {
    if (N == 0 || b_lo >= b_hi) return hipSuccess;
    const int group = ld >= 128 ? 32 : 16;
    const int HL = gatmh_sweep_hl(K, D, ld);
    SpmmArgs a{};
    a.N = N; a.F = K * D; a.ld = ld; a.xl = z; a.xg = zg; a.out = o; a.accumulate = accumulate ? 1 : 0; a.self_mode = 0;
    const int R = gatmh_sweep_rows(S, group, HL, 0);
    const GatSweepScratch c = gatmh_carve(scratch, S, N, ld, ldk);
    SweepArgs w;
    dim3 gr;
    if (!HL || !gatmh_sweep_geom(a, S, group, R, cus, b_lo, b_hi, accumulate, done, ctl, flags, c.pieces, &w, &gr)) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(done, 0, ((size_t)8 * w.nsweeps * (b_hi - b_lo) * 32 + 1) * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const dim3 bl(SWEEP_NT);
#define GFS(GRP, HLV, RR, LD) hipLaunchKernelGGL((gatmh_forward_sweep_kernel<GRP, HLV, RR, LD>), gr, bl, 0, s, a, S, w, er, a_l, c.keys, op, c.dacc, c.pos_slots, c.den_slots, K, D, ldk, el, elg)
#define GFS_R(HLV) do { if (R == 4) GFS(32, HLV, 4, true); else GFS(32, HLV, 2, true); } while (0)
#define GFS_R16(HLV) do { if (R == 4) GFS(16, HLV, 4, true); else GFS(16, HLV, 2, true); } while (0)
    if (group == 32) {
        if (R > 4) return hipErrorInvalidValue;
        if (HL == 2) GFS_R(2); else if (HL == 4) GFS_R(4); else if (HL == 8) GFS_R(8); else GFS_R(16);
    } else {
        if (R > 4) return hipErrorInvalidValue;
        if (HL == 2) GFS_R16(2); else if (HL == 4) GFS_R16(4); else if (HL == 8) GFS_R16(8); else GFS_R16(16);
    }
#undef GFS_R16
#undef GFS_R
#undef GFS
    return hipGetLastError();
}

// pieces of split rows, the self edge, the normalisation, m / den / dpos; then the rows whose denominator underflowed
hipError_t launch_gatmh_forward_sweep_finish(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const uint64_t *colptr,
                                             const uint32_t *rowidx, const BlockedAdj &S, const float *z, const float *zg, const float *el,
                                             const float *elg, const float *er, float *o, float *op, float *m, float *den, float *dpos,
                                             float *scratch, hipStream_t s) {
    if (N == 0) return hipSuccess;
    const GatSweepScratch c = gatmh_carve(scratch, S, N, ld, ldk);
    GatMhArgs a{N, K, D, ld, ldk, colptr, rowidx};
    if (S.nsplit) {
        hipLaunchKernelGGL(gatmh_sweep_combine_kernel, dim3(S.nsplit), dim3(256), 0, s, S, ld, 2 * ldk, c.pieces, c.den_slots, o, c.dacc);
        hipLaunchKernelGGL(gatmh_sweep_combine_kernel, dim3(S.nsplit), dim3(256), 0, s, S, ld, 0u, c.pos_slots, c.den_slots, op, c.dacc);
    }
    const size_t n = (size_t)N * (ld >> 2);
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(gatmh_forward_finish_kernel, dim3(blocks), dim3(256), 0, s, a, z, el, er, c.keys, c.dacc, o, op, m, den, dpos, c.redo_flag, c.redo_list);
    hipLaunchKernelGGL(gatmh_forward_redo_kernel, dim3(64), dim3(256), 0, s, a, z, zg, el, elg, er, o, op, m, den, dpos, c.redo_flag, c.redo_list);
    return hipGetLastError();
}

// destination side of the backward pass without an edge sweep: t, der, st4 from dO, O, P ("op") and dpos
hipError_t launch_gatmh_dst_rowwise(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const float *d_o, const float *o,
                                    const float *op, const float *dpos, const float *er, const float *m, const float *den, float *t,
                                    float *der, float4 *st4, uint32_t lds4, hipStream_t s) {
    if (N == 0) return hipSuccess;
    const int HL = gatmh_sweep_hl(K, D, ld);
    if (!HL || ((ld >> 2) % (uint32_t)HL) != 0) return hipErrorInvalidValue;
    GatMhArgs a{N, K, D, ld, ldk, nullptr, nullptr};
    const size_t n = (size_t)N * (ld >> 2);
    hipLaunchKernelGGL(gatmh_dst_rowwise_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, a, HL, d_o, o, op, dpos, er, m, den, t, der, st4, lds4);
    return hipGetLastError();
}

// scratch layout of one layer's source-side sweep (floats): [pieces: nslots x ld][positive-branch pieces: nslots x ld][(T, T+) slots:
// nslots x 2 ldk][S+: N x ld][(T, T+): N x 2 ldk][stx: N x K x 4][stx of the ghost destinations: G x K x 4]
size_t gatmh_src_sweep_scratch_bytes(const BlockedAdj &S, uint32_t N, uint32_t G, uint32_t K, uint32_t ld, uint32_t ldk) {
    return ((size_t)S.nslots * (2 * ld + 2 * ldk) + (size_t)N * (ld + 2 * ldk) + ((size_t)N + G) * K * 4) * sizeof(float) + 256;
}
struct GatSrcScratch {
    float *pieces, *pos_slots, *t_slots, *sp, *tacc;
    float4 *stx, *stxg;
};
static GatSrcScratch gatmh_src_carve(float *scratch, const BlockedAdj &S, uint32_t N, uint32_t K, uint32_t ld, uint32_t ldk) {
    GatSrcScratch c;
    c.pieces = scratch;
    c.pos_slots = c.pieces + (size_t)S.nslots * ld;
    c.t_slots = c.pos_slots + (size_t)S.nslots * ld;
    c.sp = c.t_slots + (size_t)S.nslots * 2 * ldk;
    c.tacc = c.sp + (size_t)N * ld;
    c.stx = reinterpret_cast<float4 *>(c.tacc + (size_t)N * 2 * ldk);      // (all pieces are multiples of 4 floats: 16-byte aligned)
    c.stxg = c.stx + (size_t)N * K;
    return c;
}

// the destinations' statistics in the form the sweep reads (local rows and, after the exchange, the ghost destinations')
hipError_t launch_gatmh_src_sweep_begin(uint32_t N, uint32_t G, uint32_t K, uint32_t ld, uint32_t ldk, const BlockedAdj &S,
                                        const float4 *st4, const float4 *stg, uint32_t lds4, float *scratch, hipStream_t s) {
    if (N == 0) return hipSuccess;
    const GatSrcScratch c = gatmh_src_carve(scratch, S, N, K, ld, ldk);
    hipLaunchKernelGGL(gatmh_stx_kernel, dim3((uint32_t)(((uint64_t)N * K + 255) / 256)), dim3(256), 0, s, (uint64_t)N * K, K, st4, lds4, c.stx);
    if (G) hipLaunchKernelGGL(gatmh_stx_kernel, dim3((uint32_t)(((uint64_t)G * K + 255) / 256)), dim3(256), 0, s, (uint64_t)G * K, K, stg, lds4, c.stxg);
    return hipGetLastError();
}

// one launch over the destination blocks [b_lo, b_hi) of the sweep layout of the OUT-edges
hipError_t launch_gatmh_src_sweep_part(uint32_t N, uint32_t G, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const BlockedAdj &S,
                                       const float *d_o, const float *dog, const float *el, float *dz, float *scratch, uint32_t cus,
                                       uint32_t b_lo, uint32_t b_hi, bool accumulate, uint32_t *done, const SweepCtl &ctl, uint32_t flags,
                                       hipStream_t s) {
    if (N == 0 || b_lo >= b_hi) return hipSuccess;
    const int group = ld >= 128 ? 32 : 16;
    const int HL = gatmh_sweep_hl(K, D, ld);
    SpmmArgs a{};
    a.N = N; a.F = K * D; a.ld = ld; a.xl = d_o; a.xg = dog; a.out = dz; a.accumulate = accumulate ? 1 : 0; a.self_mode = 0;
    const int R = gatmh_sweep_rows(S, group, HL, 1);
    const GatSrcScratch c = gatmh_src_carve(scratch, S, N, K, ld, ldk);
    SweepArgs w;
    dim3 gr;
    if (!HL || (uint64_t)(N > G ? N : G) * K * 16u >= (1ull << 32) || K * 16u >= (1u << 24) ||
        !gatmh_sweep_geom(a, S, group, R, cus, b_lo, b_hi, accumulate, done, ctl, flags, c.pieces, &w, &gr))
        return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(done, 0, ((size_t)8 * w.nsweeps * (b_hi - b_lo) * 32 + 1) * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const dim3 bl(SWEEP_NT);
#define GSS(GRP, HLV, RR, LD) hipLaunchKernelGGL((gatmh_src_sweep_kernel<GRP, HLV, RR, LD>), gr, bl, 0, s, a, S, w, el, c.stx, c.stxg, c.sp, c.tacc, c.pos_slots, c.t_slots, K, D, ldk, G)
    if (group == 32) {
        if (R != 2 && R != 4) return hipErrorInvalidValue;
#define GSS_R(HLV) do { if (R == 4) GSS(32, HLV, 4, true); else GSS(32, HLV, 2, true); } while (0)
        if (HL == 2) GSS_R(2); else if (HL == 4) GSS_R(4); else if (HL == 8) GSS_R(8); else GSS_R(16);
#undef GSS_R
    } else {
        if (R != 2 && R != 4 && R != 6) return hipErrorInvalidValue;
#define GSS_R16(HLV) do { if (R == 4) GSS(16, HLV, 4, GATMH_SRC16_LOADER); else GSS(16, HLV, 2, GATMH_SRC16_LOADER); } while (0)
        if (HL == 2) GSS_R16(2); else if (HL == 4) GSS_R16(4); else if (HL == 8) GSS_R16(8); else GSS_R16(16);
#undef GSS_R16
    }
#undef GSS
    return hipGetLastError();
}

// pieces of split rows, the self edge, del, dz
hipError_t launch_gatmh_src_sweep_finish(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const BlockedAdj &S, const float *z,
                                         const float *el, const float *d_o, const float *der, const float *a_l, const float *a_r, float *del,
                                         float *dz, float *scratch, hipStream_t s) {
    if (N == 0) return hipSuccess;
    const int HL = gatmh_sweep_hl(K, D, ld);
    if (!HL || ((ld >> 2) % (uint32_t)HL) != 0) return hipErrorInvalidValue;
    const GatSrcScratch c = gatmh_src_carve(scratch, S, N, K, ld, ldk);
    GatMhArgs a{N, K, D, ld, ldk, nullptr, nullptr};
    if (S.nsplit) {
        hipLaunchKernelGGL(gatmh_sweep_combine_kernel, dim3(S.nsplit), dim3(256), 0, s, S, ld, 2 * ldk, c.pieces, c.t_slots, dz, c.tacc);
        hipLaunchKernelGGL(gatmh_sweep_combine_kernel, dim3(S.nsplit), dim3(256), 0, s, S, ld, 0u, c.pos_slots, c.t_slots, c.sp, c.tacc);
    }
    const size_t n = (size_t)N * (ld >> 2);
    hipLaunchKernelGGL(gatmh_src_finish_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, a, HL, z, el, c.stx, d_o, c.sp, c.tacc, der,
                       a_l, a_r, dz, del);
    return hipGetLastError();
}

}  // namespace dory
