// gat_mh.hip -- multi-head GAT extension (BASELINE.json config 3 wording:
// "8-head, per-edge attention softmax + weighted SpMM").  The reference has no such code
// (its GAT is a single-head, softmax-free prototype: CPU_comm.cpp:190-242), so this is an
// extension defined by oracle/gat_mh_oracle.py (float64; parity with the reference unpinned).
//
//   el[u,k] = <Z[u,k,:], a_l[k,:]>   er[v,k] = <Z[v,k,:], a_r[k,:]>
//   s[e,k]  = LeakyReLU_0.2(el[src,k] + er[dst,k]);  alpha = softmax over in-edges of dst (+ self edge)
//   O[v,k,:] = sum_e alpha[e,k] Z[src,k,:]
//
// Nothing of size E x K is ever stored: alpha is recomputed from el/er and the per-vertex
// softmax statistics (m, den) wherever it is needed.  One wave per vertex; lane l owns
// features l, l+64, ... of the K*D-wide row (coalesced 256-B gathers); per-head reductions
// are xor-shuffles inside the D lanes that share a head.
// Backward uses the identity  der = sum a*da*l' - t * sum a*l'  (t = sum a*da) so the
// destination side needs a single sweep over the in-edges; the source side (CSR) gathers
// dO, er, m, den, t of each destination and produces del and the alpha-weighted dZ.
#include "gat_mh.hpp"

namespace dory {

// sum over the D lanes of a head group (D power of two <= 64), result in every lane of the group
__device__ __forceinline__ float head_sum(float v, int D) {
    for (int o = 1; o < D; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// el / er: one thread per (vertex, head)
__global__ void gatmh_scores_kernel(uint32_t N, uint32_t K, uint32_t D, const float *z, uint32_t ldz,
                                    const float *a_l, const float *a_r, float *el, float *er, uint32_t ldk) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)N * K) return;
    const uint32_t v = (uint32_t)(i / K), k = (uint32_t)(i % K);
    const float *zr = z + (size_t)v * ldz + (size_t)k * D;
    float sl = 0.f, sr = 0.f;
    for (uint32_t d = 0; d < D; ++d) {
        const float x = zr[d];
        sl = fmaf(x, a_l[k * D + d], sl);
        sr = fmaf(x, a_r[k * D + d], sr);
    }
    el[(size_t)v * ldk + k] = sl;
    er[(size_t)v * ldk + k] = sr;
}


// the same with one float4 of a row per thread and the G threads of a head reduced by shuffles (G = D / 4 lanes per head, or the
// whole row for a single head): coalesced 16-byte loads instead of K strided scalar walks per row (round 6: 103 -> see HISTORY)
__global__ __launch_bounds__(256) void gatmh_scores4_kernel(uint32_t N, uint32_t K, uint32_t D, uint32_t G, const float *z, uint32_t ldz,
                                                            const float *a_l, const float *a_r, float *el, float *er, uint32_t ldk) {
    const uint32_t nchunk = ldz >> 2, KD = K * D;
    const uint64_t n = (uint64_t)N * nchunk;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i < n;
    const uint64_t ii = ok ? i : 0;
    const uint32_t v = (uint32_t)(ii / nchunk), col = (uint32_t)(ii % nchunk);
    const float4 x = ok ? reinterpret_cast<const float4 *>(z)[ii] : make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t f0 = col * 4;
    float sl = 0.f, sr = 0.f;
    const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (f0 + q < KD) {
            sl = fmaf(xs[q], a_l[f0 + q], sl);
            sr = fmaf(xs[q], a_r[f0 + q], sr);
        }
    for (uint32_t off = 1; off < G; off <<= 1) {
        sl += __shfl_xor(sl, (int)off, 64);
        sr += __shfl_xor(sr, (int)off, 64);
    }
    if (ok && (col % G) == 0 && f0 < KD) {
        const uint32_t k = K == 1 ? 0u : col / G;
        el[(size_t)v * ldk + k] = sl;
        er[(size_t)v * ldk + k] = sr;
    }
}

// Forward: online softmax statistics, then alpha-weighted aggregation (self edge last).
template <int NC>
__global__ __launch_bounds__(256) void gatmh_forward_kernel(GatMhArgs a, const float *z, const float *el,
                                                            const float *er, float *o, float *m_out,
                                                            float *den_out) {
    const int lane = threadIdx.x & 63;
    const uint32_t v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= a.N) return;
    const uint32_t KD = a.K * a.D;
    // phase-1 layout: lane = j*KP + k  (j: edge inside the chunk, k: head)
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
            const float s = lrelu02(el[(size_t)u * a.ldk + k1] + er_v);
            const float mn = fmaxf(m, s);
            den = den * __expf(m - mn) + __expf(s - mn);   // m = -inf first time: exp(-inf) = 0
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
    // phase 2
    int hsel[NC];
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const uint32_t f = lane + 64 * c;
        hsel[c] = f < KD ? (int)(f / a.D) : 0;
        acc[c] = 0.f;
    }
    for (uint64_t e0 = e_beg; e0 <= e_end; e0 += EPC) {
        const uint64_t e = e0 + j1;
        uint32_t u = v;
        float alpha = 0.f;
        if (e <= e_end) {
            u = e < e_end ? a.idx[e] : v;
            if (kok) alpha = __expf(lrelu02(el[(size_t)u * a.ldk + k1] + er_v) - m) / den;
        }
        const uint32_t cnt = (uint32_t)min((uint64_t)EPC, e_end + 1 - e0);
        for (uint32_t j0 = 0; j0 < cnt; j0 += 4) {   // 4 source rows in flight; dead slots carry alpha = 0
            float zz[4][NC], al[4][NC];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t jj = min(j0 + q, cnt - 1);
                const bool live = j0 + q < cnt;
                const uint32_t uj = (uint32_t)__shfl((int)u, jj * KP, 64);
                const float *zr = z + (size_t)uj * a.ld;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const uint32_t f = lane + 64 * c;
                    const float w = __shfl(alpha, jj * KP + hsel[c], 64);
                    al[q][c] = live ? w : 0.f;
                    zz[q][c] = f < KD ? zr[f] : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = fmaf(al[q][c], zz[q][c], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const uint32_t f = lane + 64 * c;
        if (f < KD) o[(size_t)v * a.ld + f] = acc[c];
    }
}

// Backward, destination side (CSC): t[v,k] = sum a*da, der[v,k] = sum a*da*l' - t*sum a*l'
template <int NC>
__global__ __launch_bounds__(256) void gatmh_backward_dst_kernel(GatMhArgs a, const float *z, const float *el,
                                                                 const float *er, const float *m_in,
                                                                 const float *den_in, const float *d_o,
                                                                 float *t_out, float *der_out) {
    const int lane = threadIdx.x & 63;
    const uint32_t v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= a.N) return;
    const uint32_t KD = a.K * a.D;
    const int Dred = a.K == 1 ? 64 : (int)a.D;   // K == 1: the whole row is one head
    int hsel[NC];
    float dov[NC], erv[NC], mv[NC], idn[NC];
    float t[NC], a1[NC], a2[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const uint32_t f = lane + 64 * c;
        const bool ok = f < KD;
        hsel[c] = ok ? (int)(f / a.D) : 0;
        dov[c] = ok ? d_o[(size_t)v * a.ld + f] : 0.f;
        erv[c] = er[(size_t)v * a.ldk + hsel[c]];
        mv[c] = m_in[(size_t)v * a.ldk + hsel[c]];
        idn[c] = 1.f / den_in[(size_t)v * a.ldk + hsel[c]];
        t[c] = a1[c] = a2[c] = 0.f;
    }
    const uint64_t e_beg = a.ptr[v], e_end = a.ptr[v + 1];
    for (uint64_t e = e_beg; e <= e_end; ++e) {
        const uint32_t u = e < e_end ? a.idx[e] : v;
        const float *zr = z + (size_t)u * a.ld;
        float prod[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const uint32_t f = lane + 64 * c;
            prod[c] = f < KD ? dov[c] * zr[f] : 0.f;
        }
        if (a.K == 1) {   // one head spans all chunks
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) s += prod[c];
            s = head_sum(s, 64);
#pragma unroll
            for (int c = 0; c < NC; ++c) prod[c] = s;
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) prod[c] = head_sum(prod[c], Dred);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float pre = el[(size_t)u * a.ldk + hsel[c]] + erv[c];
            const float al = __expf(lrelu02(pre) - mv[c]) * idn[c];
            const float lp = pre > 0.f ? 1.f : GATMH_SLOPE;
            t[c] = fmaf(al, prod[c], t[c]);
            a1[c] = fmaf(al * prod[c], lp, a1[c]);
            a2[c] = fmaf(al, lp, a2[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const uint32_t f = lane + 64 * c;
        if (f < KD && (f % a.D) == 0) {
            t_out[(size_t)v * a.ldk + hsel[c]] = t[c];
            der_out[(size_t)v * a.ldk + hsel[c]] = a1[c] - t[c] * a2[c];
        }
    }
}

// Backward, source side (CSR): del[u,k] = sum_out dpre, dz[u,:] = sum_out alpha * dO[dst,:]
//                              + del*a_l + der*a_r  (finishing terms fused at the end)
template <int NC>
__global__ __launch_bounds__(256) void gatmh_backward_src_kernel(GatMhArgs a, const float *z, const float *el,
                                                                 const float *er, const float *m_in,
                                                                 const float *den_in, const float *t_in,
                                                                 const float *der_in, const float *d_o,
                                                                 const float *a_l, const float *a_r,
                                                                 float *del_out, float *dz) {
    const int lane = threadIdx.x & 63;
    const uint32_t u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= a.N) return;
    const uint32_t KD = a.K * a.D;
    const int Dred = a.K == 1 ? 64 : (int)a.D;
    int hsel[NC];
    float zu[NC], elu_[NC], del[NC], acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const uint32_t f = lane + 64 * c;
        const bool ok = f < KD;
        hsel[c] = ok ? (int)(f / a.D) : 0;
        zu[c] = ok ? z[(size_t)u * a.ld + f] : 0.f;
        elu_[c] = el[(size_t)u * a.ldk + hsel[c]];
        del[c] = acc[c] = 0.f;
    }
    const uint64_t e_beg = a.ptr[u], e_end = a.ptr[u + 1];
    for (uint64_t e = e_beg; e <= e_end; ++e) {
        const uint32_t v = e < e_end ? a.idx[e] : u;
        const float *dor = d_o + (size_t)v * a.ld;
        float dov[NC], prod[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const uint32_t f = lane + 64 * c;
            dov[c] = f < KD ? dor[f] : 0.f;
            prod[c] = dov[c] * zu[c];
        }
        if (a.K == 1) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) s += prod[c];
            s = head_sum(s, 64);
#pragma unroll
            for (int c = 0; c < NC; ++c) prod[c] = s;
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) prod[c] = head_sum(prod[c], Dred);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const size_t vk = (size_t)v * a.ldk + hsel[c];
            const float pre = elu_[c] + er[vk];
            const float al = __expf(lrelu02(pre) - m_in[vk]) / den_in[vk];
            const float lp = pre > 0.f ? 1.f : GATMH_SLOPE;
            del[c] = fmaf(al * (prod[c] - t_in[vk]), lp, del[c]);
            acc[c] = fmaf(al, dov[c], acc[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const uint32_t f = lane + 64 * c;
        if (f < KD) {
            const float dr = der_in[(size_t)u * a.ldk + hsel[c]];
            dz[(size_t)u * a.ld + f] = acc[c] + del[c] * a_l[f] + dr * a_r[f];
            if ((f % a.D) == 0) del_out[(size_t)u * a.ldk + hsel[c]] = del[c];
        }
    }
}

// ---- backward, (edge, head)-per-lane layout (K > 1, D in {8,16,32}) -----------------------
// lane = j*KP + k owns head k of the chunk's j-th edge and keeps that head's D-float slices
// in registers, so the per-edge dot product <dO[v,k,:], Z[u,k,:]> needs no cross-lane
// reduction (the feature-per-lane kernels above pay log2(D) shuffles per edge and chunk);
// the KP lanes of an edge read one contiguous K*D row.  Sums over edges are combined across
// the j lanes once per vertex.
template <int DL, int SP>
__global__ __launch_bounds__(256) void gatmh_backward_dst_eh_kernel(GatMhArgs a, const float *z, const float *el,
                                                                    const float *er, const float *m_in,
                                                                    const float *den_in, const float *d_o,
                                                                    float *t_out, float *der_out, float4 *st4) {
    // lane = (j*KP + k)*SP + s: edge j of the chunk, head k, s-th DL-float piece of the head's D features
    const int lane = threadIdx.x & 63;
    const uint32_t v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= a.N) return;
    uint32_t KP = 1;
    while (KP < a.K) KP <<= 1;
    const uint32_t LPE = KP * SP, EPC = 64 / LPE;
    const uint32_t sp = lane % SP, k = (lane / SP) % KP, j = lane / LPE;
    const bool kok = k < a.K;
    const uint32_t kk = kok ? k : 0;
    const uint32_t foff = kk * a.D + sp * DL;          // first feature of this lane's piece
    const uint32_t nval = sp * DL < a.D ? min((uint32_t)DL, a.D - sp * DL) : 0u;   // K = 1: D need not fill the pieces
    float dov[DL];
    {
        const float4 *p = reinterpret_cast<const float4 *>(d_o + (size_t)v * a.ld + foff);
#pragma unroll
        for (int q = 0; q < DL / 4; ++q) {
            const float4 x = p[q];
            dov[4 * q] = x.x; dov[4 * q + 1] = x.y; dov[4 * q + 2] = x.z; dov[4 * q + 3] = x.w;
        }
#pragma unroll
        for (int d = 0; d < DL; ++d) dov[d] = (uint32_t)d < nval ? dov[d] : 0.f;   // padding columns never count
    }
    const float er_v = er[(size_t)v * a.ldk + kk], m_v = m_in[(size_t)v * a.ldk + kk];
    const float idn = 1.f / den_in[(size_t)v * a.ldk + kk];
    float t = 0.f, a1 = 0.f, a2 = 0.f;
    const uint64_t e_beg = a.ptr[v], e_end = a.ptr[v + 1];
    for (uint64_t e0 = e_beg; e0 <= e_end; e0 += EPC) {
        const uint64_t e = e0 + j;
        const bool live = e <= e_end && kok;
        const uint32_t u = (e < e_end) ? a.idx[e] : v;
        const float4 *zp = reinterpret_cast<const float4 *>(z + (size_t)u * a.ld + foff);
        float da = 0.f;
#pragma unroll
        for (int q = 0; q < DL / 4; ++q) {
            const float4 x = zp[q];
            da = fmaf(dov[4 * q], x.x, da); da = fmaf(dov[4 * q + 1], x.y, da);
            da = fmaf(dov[4 * q + 2], x.z, da); da = fmaf(dov[4 * q + 3], x.w, da);
        }
#pragma unroll
        for (int o = 1; o < SP; o <<= 1) da += __shfl_xor(da, o, 64);
        const float pre = el[(size_t)u * a.ldk + kk] + er_v;
        const float al = live ? __expf(lrelu02(pre) - m_v) * idn : 0.f;
        const float lp = pre > 0.f ? 1.f : GATMH_SLOPE;
        t = fmaf(al, da, t);
        a1 = fmaf(al * da, lp, a1);
        a2 = fmaf(al, lp, a2);
    }
    for (uint32_t off = LPE; off < 64; off <<= 1) {
        t += __shfl_xor(t, off, 64);
        a1 += __shfl_xor(a1, off, 64);
        a2 += __shfl_xor(a2, off, 64);
    }
    if (j == 0 && sp == 0 && kok) {
        t_out[(size_t)v * a.ldk + k] = t;
        der_out[(size_t)v * a.ldk + k] = a1 - t * a2;
        st4[(size_t)v * a.K + k] = make_float4(er_v, m_v, idn, t);   // what the source side needs of (v,k): one 16-B gather
    }
}

template <int DL, int SP>
__global__ __launch_bounds__(256) void gatmh_backward_src_eh_kernel(GatMhArgs a, const float *z, const float *el,
                                                                    const float4 *st4, const float *der_in,
                                                                    const float *d_o, const float *a_l,
                                                                    const float *a_r, float *del_out, float *dz) {
    const int lane = threadIdx.x & 63;
    const uint32_t u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= a.N) return;
    uint32_t KP = 1;
    while (KP < a.K) KP <<= 1;
    const uint32_t LPE = KP * SP, EPC = 64 / LPE;
    const uint32_t sp = lane % SP, k = (lane / SP) % KP, j = lane / LPE;
    const bool kok = k < a.K;
    const uint32_t kk = kok ? k : 0;
    const uint32_t foff = kk * a.D + sp * DL;
    const uint32_t nval = sp * DL < a.D ? min((uint32_t)DL, a.D - sp * DL) : 0u;
    float zu[DL], acc[DL];
    {
        const float4 *p = reinterpret_cast<const float4 *>(z + (size_t)u * a.ld + foff);
#pragma unroll
        for (int q = 0; q < DL / 4; ++q) {
            const float4 x = p[q];
            zu[4 * q] = x.x; zu[4 * q + 1] = x.y; zu[4 * q + 2] = x.z; zu[4 * q + 3] = x.w;
        }
#pragma unroll
        for (int d = 0; d < DL; ++d) {
            zu[d] = (uint32_t)d < nval ? zu[d] : 0.f;
            acc[d] = 0.f;
        }
    }
    const float el_u = el[(size_t)u * a.ldk + kk];
    float del = 0.f;
    const uint64_t e_beg = a.ptr[u], e_end = a.ptr[u + 1];
    for (uint64_t e0 = e_beg; e0 <= e_end; e0 += EPC) {
        const uint64_t e = e0 + j;
        const bool live = e <= e_end && kok;
        const uint32_t v = (e < e_end) ? a.idx[e] : u;
        const float4 sv = st4[(size_t)v * a.K + kk];   // er, m, 1/den, t of (v,k)
        const float4 *dp = reinterpret_cast<const float4 *>(d_o + (size_t)v * a.ld + foff);
        float dv[DL];
        float da = 0.f;
#pragma unroll
        for (int q = 0; q < DL / 4; ++q) {
            const float4 x = dp[q];
            dv[4 * q] = x.x; dv[4 * q + 1] = x.y; dv[4 * q + 2] = x.z; dv[4 * q + 3] = x.w;
        }
#pragma unroll
        for (int d = 0; d < DL; ++d) da = fmaf(dv[d], zu[d], da);
#pragma unroll
        for (int o = 1; o < SP; o <<= 1) da += __shfl_xor(da, o, 64);
        const float pre = el_u + sv.x;
        const float al = live ? __expf(lrelu02(pre) - sv.y) * sv.z : 0.f;
        const float lp = pre > 0.f ? 1.f : GATMH_SLOPE;
        del = fmaf(al * (da - sv.w), lp, del);
#pragma unroll
        for (int d = 0; d < DL; ++d) acc[d] = fmaf(al, dv[d], acc[d]);
    }
    for (uint32_t off = LPE; off < 64; off <<= 1) {
        del += __shfl_xor(del, off, 64);
#pragma unroll
        for (int d = 0; d < DL; ++d) acc[d] += __shfl_xor(acc[d], off, 64);
    }
    if (j == 0 && kok) {
        const float dr = der_in[(size_t)u * a.ldk + k];
        float *out = dz + (size_t)u * a.ld + foff;
        const float *al_p = a_l + foff, *ar_p = a_r + foff;
#pragma unroll
        for (int d = 0; d < DL; ++d)
            if ((uint32_t)d < nval) out[d] = acc[d] + del * al_p[d] + dr * ar_p[d];
        if (sp == 0) del_out[(size_t)u * a.ldk + k] = del;
    }
}


// da_l[f] = sum_u del[u, f/D] * Z[u,f],  da_r[f] = sum_u der[u, f/D] * Z[u,f]   (two stages, deterministic; both sums in one
// pass over Z, four rows in flight per thread: the one-sum-per-launch form was a chain of dependent loads, 90 us per call)
__global__ __launch_bounds__(256) void gatmh_dattn_partial_kernel(uint32_t N, uint32_t KD, uint32_t D,
                                                                  const float *z, uint32_t ld, const float *w1, const float *w2,
                                                                  uint32_t ldk, float *partial1, float *partial2,
                                                                  uint32_t rows_per_block) {
    const uint32_t r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
    for (uint32_t f = threadIdx.x; f < KD; f += 256) {
        const uint32_t k = f / D;
        float s1 = 0.f, s2 = 0.f;
        uint32_t u = r0;
        for (; u + 4 <= r1; u += 4) {          // loads of four rows in flight, sums in row order
            float zz[4], a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                zz[i] = z[(size_t)(u + i) * ld + f];
                a[i] = w1[(size_t)(u + i) * ldk + k];
                b[i] = w2[(size_t)(u + i) * ldk + k];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { s1 = fmaf(a[i], zz[i], s1); s2 = fmaf(b[i], zz[i], s2); }
        }
        for (; u < r1; ++u) {
            const float zz = z[(size_t)u * ld + f];
            s1 = fmaf(w1[(size_t)u * ldk + k], zz, s1);
            s2 = fmaf(w2[(size_t)u * ldk + k], zz, s2);
        }
        partial1[(size_t)blockIdx.x * KD + f] = s1;
        partial2[(size_t)blockIdx.x * KD + f] = s2;
    }
}
// second stage: one wave per column, lane l adds partial rows l, l + 64, ... in that order, then a fixed butterfly over the
// lanes (deterministic; one thread per column walked all 1 024 partial rows: 44 us per call for a 128-float vector)
__global__ __launch_bounds__(256) void gatmh_colsum_final_kernel(uint32_t F, const float *partial, uint32_t nb, float *out) {
    const uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= F) return;
    float s = 0.f;
    for (uint32_t b = lane; b < nb; b += 64) s += partial[(size_t)b * F + j];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) out[j] = s;
}

// h = ELU(o); do = dh * ELU'(o); logits = mean_k o[:,k,:]; do = expand(dlogits) / K
__global__ void gatmh_elu_kernel(uint64_t rows, uint32_t cols, const float *o, uint32_t ldo, float *h, uint32_t ldh) {
    const uint64_t n = rows * cols;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / cols;
        const uint32_t c = (uint32_t)(i % cols);
        const float x = o[r * ldo + c];
        h[r * ldh + c] = x > 0.f ? x : expm1f(x);
    }
}
__global__ void gatmh_elu_bwd_kernel(uint64_t rows, uint32_t cols, const float *dh, uint32_t lddh, const float *o,
                                     uint32_t ldo, float *d_o, uint32_t lddo) {
    const uint64_t n = rows * cols;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / cols;
        const uint32_t c = (uint32_t)(i % cols);
        const float x = o[r * ldo + c];
        d_o[r * lddo + c] = dh[r * lddh + c] * (x > 0.f ? 1.f : __expf(x));
    }
}
__global__ void gatmh_head_mean_kernel(uint64_t rows, uint32_t K, uint32_t C, const float *o, uint32_t ldo,
                                       float *logits, uint32_t ldl) {
    const uint64_t n = rows * C;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / C;
        const uint32_t c = (uint32_t)(i % C);
        float s = 0.f;
        for (uint32_t k = 0; k < K; ++k) s += o[r * ldo + k * C + c];
        logits[r * ldl + c] = s / (float)K;
    }
}
__global__ void gatmh_head_expand_kernel(uint64_t rows, uint32_t K, uint32_t C, const float *dl, uint32_t lddl,
                                         float *d_o, uint32_t lddo) {
    const uint64_t n = rows * K * C;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / (K * C);
        const uint32_t c = (uint32_t)(i % C);
        d_o[r * lddo + (uint32_t)(i % (K * C))] = dl[r * lddl + c] / (float)K;
    }
}

static int grid_for(uint64_t n) { return (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); }

hipError_t launch_gatmh_scores(uint32_t N, uint32_t K, uint32_t D, const float *z, uint32_t ldz, const float *a_l,
                               const float *a_r, float *el, float *er, uint32_t ldk, hipStream_t s) {
    if (N == 0) return hipSuccess;
    // float4 form: a head spans G = D / 4 lanes (a power of two that divides the row's float4s), or one head is the whole row
    const uint32_t nchunk = ldz >> 2;
    uint32_t G = 0;
    if (!(ldz & 3) && nchunk && !(nchunk & (nchunk - 1)) && nchunk <= 64) {
        if (K == 1) G = nchunk;
        else if (!(D & 3) && !((D >> 2) & ((D >> 2) - 1)) && (D >> 2) <= 64 && nchunk % (D >> 2) == 0) G = D >> 2;
    }
    if (G) {
        const uint64_t n4 = (uint64_t)N * nchunk;
        hipLaunchKernelGGL(gatmh_scores4_kernel, dim3((uint32_t)((n4 + 255) / 256)), dim3(256), 0, s, N, K, D, G, z, ldz, a_l, a_r, el, er, ldk);
        return hipGetLastError();
    }
    const uint64_t n = (uint64_t)N * K;
    hipLaunchKernelGGL(gatmh_scores_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, N, K, D, z, ldz, a_l, a_r, el, er, ldk);
    return hipGetLastError();
}


hipError_t launch_gatmh_forward(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const uint64_t *colptr,
                                const uint32_t *rowidx, const float *z, const float *el, const float *er, float *o,
                                float *m, float *den, hipStream_t s) {
    if (N == 0) return hipSuccess;
    if (!gatmh_shape_ok(K, D)) return hipErrorInvalidValue;
    GatMhArgs a{N, K, D, ld, ldk, colptr, rowidx};
    const uint32_t KD = K * D;
    if (KD <= 64) hipLaunchKernelGGL(gatmh_forward_kernel<1>, dim3((N + 3) / 4), dim3(256), 0, s, a, z, el, er, o, m, den);
    else if (KD <= 128) hipLaunchKernelGGL(gatmh_forward_kernel<2>, dim3((N + 3) / 4), dim3(256), 0, s, a, z, el, er, o, m, den);
    else hipLaunchKernelGGL(gatmh_forward_kernel<4>, dim3((N + 3) / 4), dim3(256), 0, s, a, z, el, er, o, m, den);
    return hipGetLastError();
}

// da_l[f] = sum_u del[u, f/D] Z[u,f],  da_r likewise from der (two deterministic stages each)
hipError_t launch_gatmh_dattn(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const float *z,
                              const float *del, const float *der, float *da_l, float *da_r, float *scratch,
                              size_t scratch_bytes, hipStream_t s) {
    const uint32_t KD = K * D;
    uint32_t nb = 1024;
    while (nb > 1 && (nb > N / 64 || (size_t)2 * nb * KD * sizeof(float) > scratch_bytes)) nb >>= 1;   // >= 64 rows per block
    uint32_t rpb = (N + nb - 1) / nb;
    if (rpb == 0) rpb = 1;
    nb = (N + rpb - 1) / rpb;
    float *p1 = scratch, *p2 = scratch + (size_t)nb * KD;
    hipLaunchKernelGGL(gatmh_dattn_partial_kernel, dim3(nb), dim3(256), 0, s, N, KD, D, z, ld, del, der, ldk, p1, p2, rpb);
    hipLaunchKernelGGL(gatmh_colsum_final_kernel, dim3((KD + 3) / 4), dim3(256), 0, s, KD, p1, nb, da_l);
    hipLaunchKernelGGL(gatmh_colsum_final_kernel, dim3((KD + 3) / 4), dim3(256), 0, s, KD, p2, nb, da_r);
    return hipGetLastError();
}

hipError_t launch_gatmh_backward(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const uint64_t *colptr,
                                 const uint32_t *rowidx, const uint64_t *rowptr, const uint32_t *colidx,
                                 const float *z, const float *el, const float *er, const float *m, const float *den,
                                 const float *d_o, const float *a_l, const float *a_r, float *t, float *del,
                                 float *der, float *dz, float *da_l, float *da_r, float *scratch,
                                 size_t scratch_bytes, hipStream_t s) {
    if (N == 0) return hipSuccess;
    if (!gatmh_shape_ok(K, D)) return hipErrorInvalidValue;
    GatMhArgs ac{N, K, D, ld, ldk, colptr, rowidx};
    GatMhArgs ar{N, K, D, ld, ldk, rowptr, colidx};
    const uint32_t KD = K * D;
    const dim3 gr((N + 3) / 4), bl(256);
#define GATMH_BWD(NC)                                                                                                  \
    do {                                                                                                               \
        hipLaunchKernelGGL(gatmh_backward_dst_kernel<NC>, gr, bl, 0, s, ac, z, el, er, m, den, d_o, t, der);            \
        hipLaunchKernelGGL(gatmh_backward_src_kernel<NC>, gr, bl, 0, s, ar, z, el, er, m, den, t, der, d_o, a_l, a_r,   \
                           del, dz);                                                                                   \
    } while (0)
    // (edge, head[, piece])-per-lane kernels; st4 = (er, m, 1/den, t) per (v,k), carved from the scratch buffer
    float4 *st4 = reinterpret_cast<float4 *>(scratch);
    const size_t st4_bytes = ((size_t)N * K * sizeof(float4) + 255) & ~(size_t)255;
    const bool eh_fits = scratch_bytes >= st4_bytes + (size_t)KD * sizeof(float);
#define GATMH_BWD_EH(DL, SP)                                                                                            \
    do {                                                                                                               \
        hipLaunchKernelGGL((gatmh_backward_dst_eh_kernel<DL, SP>), gr, bl, 0, s, ac, z, el, er, m, den, d_o, t, der,    \
                           st4);                                                                                       \
        hipLaunchKernelGGL((gatmh_backward_src_eh_kernel<DL, SP>), gr, bl, 0, s, ar, z, el, st4, der, d_o, a_l, a_r,    \
                           del, dz);                                                                                   \
        scratch += st4_bytes / sizeof(float);                                                                          \
        scratch_bytes -= st4_bytes;                                                                                    \
    } while (0)
    if (eh_fits && K > 1 && D == 8) GATMH_BWD_EH(8, 1);
    else if (eh_fits && K > 1 && D == 16) GATMH_BWD_EH(16, 1);
    else if (eh_fits && K > 1 && D == 32) GATMH_BWD_EH(32, 1);
    else if (eh_fits && K == 1 && D <= 64 && ld >= 64) GATMH_BWD_EH(16, 4);   // one head split over 4 lanes (D = 41 -> 64)
    else if (eh_fits && K == 1 && D <= 128 && ld >= 128) GATMH_BWD_EH(32, 4);
    else if (KD <= 64) GATMH_BWD(1);
    else if (KD <= 128) GATMH_BWD(2);
    else GATMH_BWD(4);
#undef GATMH_BWD
#undef GATMH_BWD_EH
    return launch_gatmh_dattn(N, K, D, ld, ldk, z, del, der, da_l, da_r, scratch, scratch_bytes, s);
}

hipError_t launch_gatmh_elu(uint64_t rows, uint32_t cols, const float *o, uint32_t ldo, float *h, uint32_t ldh, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(gatmh_elu_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, s, rows, cols, o, ldo, h, ldh);
    return hipGetLastError();
}
hipError_t launch_gatmh_elu_bwd(uint64_t rows, uint32_t cols, const float *dh, uint32_t lddh, const float *o,
                                uint32_t ldo, float *d_o, uint32_t lddo, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(gatmh_elu_bwd_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, s, rows, cols, dh, lddh, o, ldo, d_o, lddo);
    return hipGetLastError();
}
hipError_t launch_gatmh_head_mean(uint64_t rows, uint32_t K, uint32_t C, const float *o, uint32_t ldo, float *logits,
                                  uint32_t ldl, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(gatmh_head_mean_kernel, dim3(grid_for(rows * C)), dim3(256), 0, s, rows, K, C, o, ldo, logits, ldl);
    return hipGetLastError();
}
hipError_t launch_gatmh_head_expand(uint64_t rows, uint32_t K, uint32_t C, const float *dl, uint32_t lddl, float *d_o,
                                    uint32_t lddo, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(gatmh_head_expand_kernel, dim3(grid_for(rows * K * C)), dim3(256), 0, s, rows, K, C, dl, lddl, d_o, lddo);
    return hipGetLastError();
}

}  // namespace dory
