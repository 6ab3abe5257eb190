// gemm.hip -- K2: fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32,
// exact fp32 = a k-ordered fmaf chain), with the tanh epilogue (K3) fused.
//
// Replaces Matrix::dot -> cblas_sgemm (reference src/common/matrix.cpp:263-315) at
// its call sites in CPUComm::vtxNN{Forward,Backward}{GCN,GAT}
// (src/graph-server/commmanager/CPU_comm.cpp:98-188) and the cuBLAS path
// CuMatrix::dot (GPU-Computation/cu_matrix.cu:205-232) + cudnnActivationForward
// (comp_unit.cu:241-256).
//
// Shapes on this path are tall-skinny: M = |V_local| (2.3e5 .. 8e6), K,N <= 1433.
//   NN  z    = ah * W          (+ h = tanh(z) epilogue)
//   NT  grad = d  * W^T
//   TN  dW   = ah^T * d        reduction over M: split-K over workgroups with a
//                              deterministic second-stage sum (no atomics)
// Tiling: 128 x BN block (BN = 128 or 64), 16-deep k-tiles, 4 waves, each wave a 64x64 (2x2 MFMA tiles) or 32x64 (1x2)
// register tile; operand tiles travel global -> LDS by LDS-DMA into two stages (below).  Sum order per output element: k-tiles
// in order, inside a k-tile the pairs (m, 8+m) for m = 0..7 -- fixed, whatever the grid.
#include "ctx.hpp"

namespace dory {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// k-tile depth: 16 (two 16-KB stages per 128-wide workgroup -> 4 workgroups per CU)
constexpr int BK_SPLIT = 32;   // granularity of the split-K plan

// ---------------------------------------------------------------------------------------------------------------------
// The body (round 5; the register-staged form it replaces -- bounds-checked loads in branches, a transposing pass of
// ds_write_b32, one fragment register set -- measured 372 / 332 us for the 602->128 NN / TN launches inside the Reddit epoch,
// this one 348 / 303 us; profiles/r05_gemm_dma_ab.txt).  Both operand tiles travel global -> LDS by `buffer_load_dwordx4 ... lds` (1 KB per wave instruction, no
// VGPRs, no ds_write), out-of-range rows / k read zeros through the buffer resource (no branches), and the k-tile's fragments
// are all read before its 8 x TM x TN MFMAs.  LDS images (the DMA writes lane-linear, so any swizzle is on the SOURCE side):
//   k-major source  elem(k,i) = p[k*ld+i]  ->  image[k][i] as it lies; fragment = 8 ds_read_b32 (rows 8*fk+m, 32 consecutive i)
//   row-major source elem(k,i) = p[i*ld+k] ->  image[i][16 k] with the row's four 16-byte quads XOR-ed by (i>>2)&3; fragment =
//                                              two ds_read_b128 (quads 2*fk, 2*fk+1): conflict-free in ds_read_b128's lane groups
// MFMA m of a k-tile multiplies k = m (lanes 0-31) and k = 8+m (lanes 32-63) -- the same pairing for both image kinds.
template <int BT, bool KMAJOR, int BK>
struct DmaTile {
    static_assert(BK == 16, "image geometry is written for 16-deep k-tiles");
    static constexpr int PIECES = BT * BK * 4 / 1024 / 4;   // 1-KB pieces per wave (4 waves)
    uint32_t voff;       // this lane's byte offset inside a piece's source (constant over the k-loop)
    uint32_t pstride;    // byte distance between this wave's consecutive pieces
    uint32_t step;       // bytes the source advances per k-tile
    const char *base;    // source of the current k-tile
    int64_t left;        // bytes readable from base on (rows / k beyond it read zeros)

    __device__ __forceinline__ void init(const float *p, uint32_t ld, uint32_t ext, uint32_t i0, uint32_t k0, uint32_t kend) {
        const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const uint64_t row_b = (uint64_t)ld * 4u;
        if constexpr (KMAJOR) {   // a piece = 1024 / (BT*4) whole k-rows of the tile
            constexpr uint32_t KPP = 256 / BT, LPR = BT / 4;
            base = reinterpret_cast<const char *>(p) + (uint64_t)k0 * row_b + (uint64_t)i0 * 4u;
            voff = (wave * PIECES * KPP + lane / LPR) * (uint32_t)row_b + (lane % LPR) * 16u;
            pstride = KPP * (uint32_t)row_b;
            step = BK * (uint32_t)row_b;
            left = (int64_t)(kend - k0) * (int64_t)row_b - (int64_t)i0 * 4;
        } else {                  // a piece = 16 rows x 64 bytes; quads swizzled by the row
            const uint32_t r = wave * PIECES * 16u + (lane >> 2);
            const uint32_t q = (lane & 3u) ^ ((r >> 2) & 3u);
            base = reinterpret_cast<const char *>(p) + (uint64_t)i0 * row_b + (uint64_t)k0 * 4u;
            voff = r * (uint32_t)row_b + q * 16u;
            pstride = 16u * (uint32_t)row_b;
            step = BK * 4u;
            left = (int64_t)(ext - i0) * (int64_t)row_b - (int64_t)k0 * 4;
        }
    }
    // issue this wave's pieces of the current k-tile into `img` (BT*BK floats), then move on to the next k-tile
    __device__ __forceinline__ void issue(float *img) {
        const uint32_t wave = threadIdx.x >> 6;
        const uint32_t n = left <= 0 ? 0u : (left > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)left);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, n, 0x00020000);
        typedef __attribute__((address_space(3))) void *lptr_t;
#pragma unroll
        for (int pc = 0; pc < PIECES; ++pc)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(img + (wave * PIECES + pc) * 256), 16, voff, pc * pstride, 0, 0);
        base += step;
        left -= step;
    }
};

// the eight k-values of a k-tile this lane feeds to the MFMAs, for the 32-row block `blk` of an image
template <int BT, bool KMAJOR>
__device__ __forceinline__ void dma_frag(float (&f)[8], const float *img, int blk, int fr, int fk) {
    if constexpr (KMAJOR) {
#pragma unroll
        for (int m = 0; m < 8; ++m) f[m] = img[(8 * fk + m) * BT + blk * 32 + fr];
    } else {
        const int r = blk * 32 + fr;
        const int sw = (r >> 2) & 3;
        const float4 lo = *reinterpret_cast<const float4 *>(img + r * 16 + ((2 * fk) ^ sw) * 4);
        const float4 hi = *reinterpret_cast<const float4 *>(img + r * 16 + ((2 * fk + 1) ^ sw) * 4);
        f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w;
        f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
    }
}

template <int BN, int WM, int WN, int TM, int TN, bool A_KMAJOR, bool B_KMAJOR, int BK>
__device__ __forceinline__ void gemm_dma_body(const GemmArgs &g, uint32_t klen, float *partial, float *smem) {
    static_assert(WM * WN == 4 && WN * TN * 32 == BN, "tile shape");
    constexpr int BM = WM * TM * 32;
    constexpr int A_SZ = BM * BK, B_SZ = BN * BK, STAGE = A_SZ + B_SZ;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const uint32_t i0 = blockIdx.x * BM;
    const uint32_t j0 = blockIdx.y * BN;
    const uint32_t kbeg = blockIdx.z * klen;
    const uint32_t kend = min(g.K, kbeg + klen);
    const int fr = lane & 31, fk = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    DmaTile<BM, A_KMAJOR, BK> ta;
    DmaTile<BN, B_KMAJOR, BK> tb;
    ta.init(g.A, g.lda, g.M, i0, kbeg, kend);
    tb.init(g.B, g.ldb, g.N, j0, kbeg, kend);
    const uint32_t T = kbeg < kend ? (kend - kbeg + BK - 1) / BK : 0;
    // (measured and dropped, round 5: starting the first resident set apart -- by launch index, then by arrival order on the CU
    // (HW_ID), an eighth / quarter / half of a workgroup's duration per slot -- so that tile write-backs and first loads of one
    // workgroup fall under another's MFMAs: 345 -> 348-357 us.  One resident round (M = 131 072) runs at the same fraction of
    // peak as 1.8 rounds, so workgroup hand-over is not the cost either; what K = 608 loses against K = 4 864 is per tile)
    if (T) { ta.issue(smem); tb.issue(smem + A_SZ); }
    for (uint32_t t = 0; t < T; ++t) {
        // this wave's pieces of tile t have landed; past the barrier everybody's have, and nobody still reads the other stage
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float *As = smem + (t & 1) * STAGE;
        const float *Bs = As + A_SZ;
        float af[TM][8], bf[TN][8];
#pragma unroll
        for (int a = 0; a < TM; ++a) dma_frag<BM, A_KMAJOR>(af[a], As, wm * TM + a, fr, fk);
#pragma unroll
        for (int b = 0; b < TN; ++b) dma_frag<BN, B_KMAJOR>(bf[b], Bs, wn * TN + b, fr, fk);
        // the next tile's DMA goes out AFTER this tile's fragment reads: hipcc orders any LDS read behind a pending LDS-DMA
        // with vmcnt(0), which would put the DMA's latency in front of the MFMAs instead of under them
        if (t + 1 < T) {
            float *nx = smem + ((t + 1) & 1) * STAGE;
            ta.issue(nx);
            tb.issue(nx + A_SZ);
        }
        if (t + 1 == T && (kend & (BK - 1))) {   // a row-major operand's last quad may hold k >= K: whatever lies there must not count
            const uint32_t kb = kbeg + t * BK + 8 * fk;
#pragma unroll
            for (int m = 0; m < 8; ++m)
                if (kb + m >= kend) {
#pragma unroll
                    for (int a = 0; a < TM; ++a) af[a][m] = 0.f;
#pragma unroll
                    for (int b = 0; b < TN; ++b) bf[b][m] = 0.f;
                }
        }
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][m], bf[b][m], acc[a][b], 0, 0, 0);
    }

    const bool split = gridDim.z > 1;
    float *C = split ? partial + (size_t)blockIdx.z * g.M * g.ldc : g.C;
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const uint32_t col = j0 + (wn * TN + b) * 32 + fr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t row = i0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
                if (row < g.M && col < g.N) {
                    const float v = acc[a][b][r];
                    C[(size_t)row * g.ldc + col] = v;
                    if (!split && g.epilogue == EPI_TANH)
                        g.C2[(size_t)row * g.ldc2 + col] = dory_tanh(v);
                }
            }
        }
}

template <int BN, int WM, int WN, int TM, int TN, bool A_KMAJOR, bool B_KMAJOR, int BK>
__global__ __launch_bounds__(256) void gemm_dma_kernel(GemmArgs g, uint32_t klen, float *partial) {
    __shared__ __attribute__((aligned(1024))) float smem[2 * (WM * TM * 32 + BN) * BK];
    gemm_dma_body<BN, WM, WN, TM, TN, A_KMAJOR, B_KMAJOR, BK>(g, klen, partial, smem);
}
template <int BN, int WM, int WN, int TM, int TN, bool A_KMAJOR, bool B_KMAJOR, int BK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_dma_kernel_occ4(GemmArgs g, uint32_t klen,
                                                                                                     float *partial) {
    __shared__ __attribute__((aligned(1024))) float smem[2 * (WM * TM * 32 + BN) * BK];
    gemm_dma_body<BN, WM, WN, TM, TN, A_KMAJOR, B_KMAJOR, BK>(g, klen, partial, smem);
}

// second stage of split-K: C = sum_z partial[z] in z order (deterministic); eight
// partials are requested at a time so the loads overlap, the adds stay sequential
__global__ void splitk_reduce_kernel(float *C, const float *partial, uint32_t M, uint32_t N,
                                     uint32_t ldc, uint32_t S, float *C2 /*EPI_TANH: tanh(C), or nullptr*/,
                                     uint32_t ldc2) {
    const size_t n = (size_t)M * ldc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        if ((i % ldc) >= N) continue;
        float s = 0.f;
        uint32_t z = 0;
        for (; z + 8 <= S; z += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(z + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < S; ++z) s += partial[(size_t)z * n + i];
        C[i] = s;
        if (C2) C2[(i / ldc) * ldc2 + (i % ldc)] = dory_tanh(s);
    }
}

#ifndef GEMM_BM_WIDE
#define GEMM_BM_WIDE 128   // rows per tile of the 128-wide shape; 64 (finer tail, 6 workgroups per CU) measured: NN 344 -> 339 us, TN 302 -> 348, NT 50 -> 44
#endif
#ifndef GEMM_BM_NARROW
#define GEMM_BM_NARROW 128   // rows per tile of the 64-wide shape (N <= 64); 256 (every wave a 64 x 64 register tile, 140 VGPRs) measured round 6:
                             // Amazon 300 -> 64: NN 602 -> 595-602 us, TN 544-551 -> 687-691; Reddit 128 -> 41: NN 61 -> 64, TN 85 -> 113 us
#endif
static uint32_t pick_splits(uint32_t M, uint32_t N, uint32_t K, int bm, int bn) {
    const uint32_t tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    if (tiles == 0) return 1;
    // a workgroup walks its k-tiles one after the other (~1.9 us each): with few tiles even a
    // Cora-sized K (2 708 rows -> 85 k-tiles = 160 us on one workgroup) wants to be split
    if (tiles >= 512 || K < 8 * BK_SPLIT) return 1;
    uint32_t s = (1024 + tiles - 1) / tiles;            // aim at ~1024 workgroups = 256 CUs x 4 resident blocks
    const uint32_t maxs = (K + 4 * BK_SPLIT - 1) / (4 * BK_SPLIT);  // at least 128 k-rows per split
    if (s > maxs) s = maxs;
    if (s > 512) s = 512;
    return s < 1 ? 1 : s;
}

size_t gemm_scratch_bytes(uint32_t M, uint32_t N, uint32_t K) {
    // only split-K (few output tiles, long K) uses scratch: one M x ld(N) partial per split
    const uint32_t S = pick_splits(M, N, K, N > 64 ? GEMM_BM_WIDE : GEMM_BM_NARROW, N > 64 ? 128 : 64);   // (the two shapes of launch_gemm)
    return S > 1 ? (size_t)S * M * pad_ld(N) * sizeof(float) : 0;
}

template <int BN, int WM, int WN, int TM, int TN, int BK>
static hipError_t launch_bn(const GemmArgs &g, float *scratch, size_t scratch_bytes, hipStream_t s) {
    if (g.K == 0) {   // empty partition: the product of an M x 0 and a 0 x N matrix is all zeros (tanh(0) = 0 too)
        hipError_t e = hipMemsetAsync(g.C, 0, (size_t)g.M * g.ldc * sizeof(float), s);
        if (e == hipSuccess && g.epilogue == EPI_TANH && g.C2)
            e = hipMemsetAsync(g.C2, 0, (size_t)g.M * g.ldc2 * sizeof(float), s);
        return e;
    }
    constexpr int BM = WM * TM * 32;
    uint32_t S = pick_splits(g.M, g.N, g.K, BM, BN);
    if (S > 1 && (size_t)S * g.M * g.ldc * sizeof(float) > scratch_bytes) {
        S = (uint32_t)(scratch_bytes / ((size_t)g.M * g.ldc * sizeof(float)));
        if (S < 2) S = 1;
    }
    uint32_t klen = (g.K + S - 1) / S;
    klen = (klen + BK - 1) / BK * BK;
    S = (g.K + klen - 1) / klen;
    dim3 grid((g.M + BM - 1) / BM, (g.N + BN - 1) / BN, S);
    dim3 block(256);
#define GEMM_LAUNCH(AK, BKM)                                                                                                  \
    do {                                                                                                                      \
        if constexpr (BN == 128 && WM * TM * 32 == 128)                                                                       \
            hipLaunchKernelGGL((gemm_dma_kernel_occ4<BN, WM, WN, TM, TN, AK, BKM, BK>), grid, block, 0, s, g, klen, scratch); \
        else                                                                                                                  \
            hipLaunchKernelGGL((gemm_dma_kernel<BN, WM, WN, TM, TN, AK, BKM, BK>), grid, block, 0, s, g, klen, scratch);      \
    } while (0)
    if (!g.ta && !g.tb) GEMM_LAUNCH(false, true);
    else if (!g.ta && g.tb) GEMM_LAUNCH(false, false);
    else if (g.ta && !g.tb) GEMM_LAUNCH(true, true);
    else return hipErrorInvalidValue;
#undef GEMM_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (S > 1) {
        const size_t n = (size_t)g.M * g.ldc;
        int blocks = (int)((n + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, g.C, scratch, g.M, g.N, g.ldc, S,
                           g.epilogue == EPI_TANH ? g.C2 : (float *)nullptr, g.ldc2);
        e = hipGetLastError();
    }
    return e;
}

hipError_t launch_gemm(const GemmArgs &g, float *scratch, size_t scratch_bytes, hipStream_t s) {
    if (g.M == 0 || g.N == 0) return hipSuccess;
    // the DMA moves 16-byte pieces: rows must start on 16-byte boundaries (every tensor of the table does: ld is padded to 32 floats)
    if ((g.lda & 3u) || (g.ldb & 3u) || (reinterpret_cast<uintptr_t>(g.A) & 15u) || (reinterpret_cast<uintptr_t>(g.B) & 15u))
        return hipErrorInvalidValue;
    if (g.N > 64) return launch_bn<128, 2, 2, GEMM_BM_WIDE / 64, 2, 16>(g, scratch, scratch_bytes, s);
    return launch_bn<64, 4, 1, GEMM_BM_NARROW / 128, 2, 16>(g, scratch, scratch_bytes, s);
}

}  // namespace dory
