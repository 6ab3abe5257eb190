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
// Tiling: 128 x BN block (BN = 128 or 64), BK = 16, 4 waves, each wave a
// 64x64 (2x2 MFMA tiles) or 32x64 (1x2) register tile.  Both operands are staged
// k-major in LDS (As[k][i], Bs[k][j]) so every fragment read is a conflict-free
// ds_read_b32 of 32 consecutive floats per half-wave.
#include "ctx.hpp"

namespace dory {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
// k-tile depth is a template parameter of the kernels below; 16 everywhere: 34 KB of LDS per 128-wide workgroup -> 4
// workgroups per CU (with 32 only 2 fit and the MFMA-bound 602->128 GEMMs run 7 % slower); the HBM-bound 64-wide
// GEMMs of an Amazon-sized partition measured the same or a few per cent better with 16 than with 32
constexpr int BK_SPLIT = 32;   // granularity of the split-K plan

// A (BK x BT) operand tile travels global -> registers -> LDS (k-major).  The two halves
// are separate so that the loads of tile t+1 are in flight while tile t is multiplied.
// kmajor source: elem(k,i) = p[k*ld+i]; otherwise elem(k,i) = p[i*ld+k] (transposing copy).
template <int BT, bool KMAJOR, int BK>
struct TileRegs {
    static constexpr int N4 = BT * BK / 4 / 256;   // float4 per thread
    float4 v[N4];
};

template <int BT, bool KMAJOR, int BK>
__device__ __forceinline__ void tile_load(TileRegs<BT, KMAJOR, BK> &r, const float *__restrict__ p, uint32_t ld,
                                          uint32_t ext, uint32_t Kend, uint32_t i0, uint32_t k0) {
    const int t = threadIdx.x;
    if constexpr (KMAJOR) {
        constexpr int VPR = BT / 4;          // float4 per k-row
        constexpr int KPI = 256 / VPR;       // k rows per iteration
        const int i4 = (t % VPR) * 4;
#pragma unroll
        for (int it = 0; it < TileRegs<BT, KMAJOR, BK>::N4; ++it) {
            const uint32_t k = k0 + t / VPR + it * KPI;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < Kend) {
                const float *src = p + (size_t)k * ld + i0 + i4;
                if (i0 + i4 + 3 < ext) {
                    v = *reinterpret_cast<const float4 *>(src);
                } else {
                    if (i0 + i4 + 0 < ext) v.x = src[0];
                    if (i0 + i4 + 1 < ext) v.y = src[1];
                    if (i0 + i4 + 2 < ext) v.z = src[2];
                }
            }
            r.v[it] = v;
        }
    } else {
        constexpr int LPR = BK / 4;          // lanes per row (8)
        constexpr int RPI = 256 / LPR;       // rows per iteration (32)
        const int kq = (t % LPR) * 4;
#pragma unroll
        for (int it = 0; it < TileRegs<BT, KMAJOR, BK>::N4; ++it) {
            const uint32_t i = i0 + t / LPR + it * RPI;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < ext) {
                const float *src = p + (size_t)i * ld + k0 + kq;
                if (k0 + kq + 3 < Kend) {
                    v = *reinterpret_cast<const float4 *>(src);
                } else {
                    if (k0 + kq + 0 < Kend) v.x = src[0];
                    if (k0 + kq + 1 < Kend) v.y = src[1];
                    if (k0 + kq + 2 < Kend) v.z = src[2];
                }
            }
            r.v[it] = v;
        }
    }
}

template <int BT, bool KMAJOR, int LLD, int BK>
__device__ __forceinline__ void tile_store(float *lds, const TileRegs<BT, KMAJOR, BK> &r) {
    const int t = threadIdx.x;
    if constexpr (KMAJOR) {
        constexpr int VPR = BT / 4;
        constexpr int KPI = 256 / VPR;
        const int i4 = (t % VPR) * 4;
#pragma unroll
        for (int it = 0; it < TileRegs<BT, KMAJOR, BK>::N4; ++it)
            *reinterpret_cast<float4 *>(lds + (t / VPR + it * KPI) * LLD + i4) = r.v[it];
    } else {
        constexpr int LPR = BK / 4;
        constexpr int RPI = 256 / LPR;
        const int kq = (t % LPR) * 4;
#pragma unroll
        for (int it = 0; it < TileRegs<BT, KMAJOR, BK>::N4; ++it) {
            const int rr = t / LPR + it * RPI;
            lds[(kq + 0) * LLD + rr] = r.v[it].x;
            lds[(kq + 1) * LLD + rr] = r.v[it].y;
            lds[(kq + 2) * LLD + rr] = r.v[it].z;
            lds[(kq + 3) * LLD + rr] = r.v[it].w;
        }
    }
}

// LDS geometry of one (BN, operand layouts, BK) instantiation.  Row-major operands are transposed on their way into
// LDS: BK/4 k-lanes x 32/(BK/4) rows per 32-lane group write (kq + c) * LD + rr, so the row pitch is chosen to keep
// the 32 banks distinct: LD = 2 (mod 32) for BK = 16 (kq in {0,4,8,12}, rr in 0..7), LD = 1 (mod 32) for BK = 32.
template <int BN, bool A_KMAJOR, bool B_KMAJOR, int BK>
struct GemmLds {
    static constexpr int TPAD = BK == 16 ? 2 : 1;
    static constexpr int LDA_S = A_KMAJOR ? BM + 4 : BM + TPAD;
    static constexpr int LDB_S = B_KMAJOR ? BN + 4 : BN + TPAD;
    static constexpr int A_SZ = (BK * LDA_S + 3) & ~3;
    static constexpr int B_SZ = (BK * LDB_S + 3) & ~3;
    static constexpr int FLOATS = 2 * (A_SZ + B_SZ);   // two buffers per operand: tile t+1 is written while nobody reads it
};

template <int BN, int WM, int WN, int TM, int TN, bool A_KMAJOR, bool B_KMAJOR, int BK>
__device__ __forceinline__ void gemm_body(const GemmArgs &g, uint32_t klen, float *partial, float *smem) {
    static_assert(WM * WN == 4 && WM * TM * 32 == BM && WN * TN * 32 == BN, "tile shape");
    using Lds = GemmLds<BN, A_KMAJOR, B_KMAJOR, BK>;
    constexpr int LDA_S = Lds::LDA_S, LDB_S = Lds::LDB_S, A_SZ = Lds::A_SZ, B_SZ = Lds::B_SZ;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const uint32_t i0 = blockIdx.x * BM;
    const uint32_t j0 = blockIdx.y * BN;
    const uint32_t kbeg = blockIdx.z * klen;
    const uint32_t kend = min(g.K, kbeg + klen);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int fr = lane & 31;   // row/col inside the 32-wide fragment
    const int fk = lane >> 5;   // k offset inside the 2-deep MFMA
    TileRegs<BM, A_KMAJOR, BK> ra;
    TileRegs<BN, B_KMAJOR, BK> rb;
    int cur = 0;
    if (kbeg < kend) {
        tile_load<BM, A_KMAJOR, BK>(ra, g.A, g.lda, g.M, kend, i0, kbeg);
        tile_load<BN, B_KMAJOR, BK>(rb, g.B, g.ldb, g.N, kend, j0, kbeg);
        tile_store<BM, A_KMAJOR, LDA_S, BK>(smem, ra);
        tile_store<BN, B_KMAJOR, LDB_S, BK>(smem + A_SZ, rb);
    }
    __syncthreads();
    for (uint32_t k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) {  // next tile: global -> registers, in flight during the MFMAs below
            tile_load<BM, A_KMAJOR, BK>(ra, g.A, g.lda, g.M, kend, i0, k0 + BK);
            tile_load<BN, B_KMAJOR, BK>(rb, g.B, g.ldb, g.N, kend, j0, k0 + BK);
        }
        const float *As = smem + cur * (A_SZ + B_SZ);
        const float *Bs = As + A_SZ;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = As[(kk + fk) * LDA_S + (wm * TM + a) * 32 + fr];
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = Bs[(kk + fk) * LDB_S + (wn * TN + b) * 32 + fr];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (more) {
            float *An = smem + (cur ^ 1) * (A_SZ + B_SZ);
            tile_store<BM, A_KMAJOR, LDA_S, BK>(An, ra);
            tile_store<BN, B_KMAJOR, LDB_S, BK>(An + A_SZ, rb);
        }
        __syncthreads();
        cur ^= 1;
    }

    // C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
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

// 64-wide tiles: the registers fit six waves per SIMD without a hint
template <int BN, int WM, int WN, int TM, int TN, bool A_KMAJOR, bool B_KMAJOR, int BK>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g, uint32_t klen, float *partial) {
    __shared__ __attribute__((aligned(16))) float smem[GemmLds<BN, A_KMAJOR, B_KMAJOR, BK>::FLOATS];
    gemm_body<BN, WM, WN, TM, TN, A_KMAJOR, B_KMAJOR, BK>(g, klen, partial, smem);
}
// 128-wide tiles with BK = 16: 34 KB of LDS -> four workgroups per CU if the kernel stays within 128 registers
template <int BN, int WM, int WN, int TM, int TN, bool A_KMAJOR, bool B_KMAJOR, int BK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_kernel_occ4(GemmArgs g, uint32_t klen,
                                                                                                 float *partial) {
    __shared__ __attribute__((aligned(16))) float smem[GemmLds<BN, A_KMAJOR, B_KMAJOR, BK>::FLOATS];
    gemm_body<BN, WM, WN, TM, TN, A_KMAJOR, B_KMAJOR, BK>(g, klen, partial, smem);
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

static uint32_t pick_splits(uint32_t M, uint32_t N, uint32_t K, int bn) {
    const uint32_t tiles = ((M + BM - 1) / BM) * ((N + bn - 1) / bn);
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
    const uint32_t S = pick_splits(M, N, K, N > 64 ? 128 : 64);
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
    uint32_t S = pick_splits(g.M, g.N, g.K, BN);
    if (S > 1 && (size_t)S * g.M * g.ldc * sizeof(float) > scratch_bytes) {
        S = (uint32_t)(scratch_bytes / ((size_t)g.M * g.ldc * sizeof(float)));
        if (S < 2) S = 1;
    }
    uint32_t klen = (g.K + S - 1) / S;
    klen = (klen + BK - 1) / BK * BK;
    S = (g.K + klen - 1) / klen;
    dim3 grid((g.M + BM - 1) / BM, (g.N + BN - 1) / BN, S);
    dim3 block(256);
#define GEMM_LAUNCH(AK, BKM)                                                                                              \
    do {                                                                                                                  \
        if constexpr (BN == 128)                                                                                          \
            hipLaunchKernelGGL((gemm_kernel_occ4<BN, WM, WN, TM, TN, AK, BKM, BK>), grid, block, 0, s, g, klen, scratch); \
        else                                                                                                              \
            hipLaunchKernelGGL((gemm_kernel<BN, WM, WN, TM, TN, AK, BKM, BK>), grid, block, 0, s, g, klen, scratch);      \
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
    if (g.N > 64) return launch_bn<128, 2, 2, 2, 2, 16>(g, scratch, scratch_bytes, s);
    return launch_bn<64, 4, 1, 1, 2, 16>(g, scratch, scratch_bytes, s);
}

}  // namespace dory
