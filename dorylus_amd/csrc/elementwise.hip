// elementwise.hip -- K3/K4/K5/K6/K7: the HBM-bound helpers around the two big
// kernels.  All are streaming kernels (16-B lanes where the layout allows),
// fp32, gfx950.  Reference call sites are cited per kernel; paths are relative
// to the reference's src/graph-server/ unless they start with src/.
#include "ctx.hpp"

namespace dory {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- K3: g = aTg * (1 - tanh(z)^2)  ------------------------------------------
// activateDerivative + Matrix::operator* in vtxNNBackwardGCN
// (commmanager/CPU_comm.cpp:140-143,436-446); cudnnActivationBackward in the
// CUDA backend (GPU-Computation/comp_unit.cu:213-239).
__global__ void tanh_backward_kernel(uint64_t rows, uint32_t cols, const float *aTg, uint32_t lda,
                                     const float *z, uint32_t ldz, float *g, uint32_t ldg) {
    const uint32_t c4 = (cols + 3) / 4;
    const uint64_t n = rows * c4;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / c4;
        const uint32_t c = (uint32_t)(i % c4) * 4;
        const float4 a = *reinterpret_cast<const float4 *>(aTg + r * lda + c);
        const float4 zz = *reinterpret_cast<const float4 *>(z + r * ldz + c);
        float4 o;
        float t;
        t = dory_tanh(zz.x); o.x = a.x * (1.f - t * t);
        t = dory_tanh(zz.y); o.y = a.y * (1.f - t * t);
        t = dory_tanh(zz.z); o.z = a.z * (1.f - t * t);
        t = dory_tanh(zz.w); o.w = a.w * (1.f - t * t);
        // padding columns of aTg are zero, so padding of g stays zero
        *reinterpret_cast<float4 *>(g + r * ldg + c) = o;
    }
}

hipError_t launch_tanh_backward(uint64_t rows, uint32_t cols, const float *aTg, uint32_t lda,
                                const float *z, uint32_t ldz, float *g, uint32_t ldg, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    const uint64_t n = rows * ((cols + 3) / 4);
    int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(tanh_backward_kernel, dim3(blocks), dim3(256), 0, s, rows, cols, aTg, lda, z, ldz, g, ldg);
    return hipGetLastError();
}

// h = tanh(z) on its own: the transform-first order computes z with the SpMM, not the GEMM whose
// epilogue normally applies the activation (same tanhf as that epilogue)
__global__ void tanh_forward_kernel(uint64_t rows, uint32_t cols, const float *z, uint32_t ldz, float *h, uint32_t ldh) {
    const uint64_t n = rows * cols;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / cols;
        const uint32_t c = (uint32_t)(i % cols);
        h[r * ldh + c] = dory_tanh(z[r * ldz + c]);
    }
}
hipError_t launch_tanh_forward(uint64_t rows, uint32_t cols, const float *z, uint32_t ldz, float *h, uint32_t ldh,
                               hipStream_t s) {
    if (rows == 0 || cols == 0) return hipSuccess;
    const uint64_t n = rows * cols;
    int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(tanh_forward_kernel, dim3(blocks), dim3(256), 0, s, rows, cols, z, ldz, h, ldh);
    return hipGetLastError();
}

// ---- K4: softmax + maskout quirk + (p - lab) / denom  ---------------------------
// softmax (CPU_comm.cpp:276-297: max-subtracted, denominator seeded with 1e-20),
// maskout (CPU_comm.cpp:464-471: copies (rows - stt) FLOATS of the dense label
// tensor over the dense prediction tensor starting at row stt -- reproduced as a
// flat dense-index range), hadamardSub + "/= globalVtxCnt * TRAIN_PORTION"
// (CPU_comm.cpp:121-122).  CUDA backend: cudnnSoftmaxForward + thrust minus +
// cublasSscal + cudaMemcpy maskout (comp_unit.cu:161-210,331-346).
// LPR lanes per row (64, 32, 16 or 8: the smallest that holds a row of up to 64 / 32 / 16 / 8 classes, so a wave
// covers 1-8 rows and no lane idles on narrow label sets); lane c owns columns c, c+LPR, ...  The xor-shuffle trees
// descend from LPR/2, i.e. the sums are the ones the 64-lane tree gives with the unused lanes at zero.
template <int LPR>
__global__ __launch_bounds__(256) void softmax_xent_kernel(
    uint32_t rows, uint32_t cols, const float *z, uint32_t ldz, const float *lab, uint32_t ldl,
    float *d, uint32_t ldd, float inv_mode_denom, uint64_t mask_first, uint64_t mask_count,
    int sub_only) {
    const int lane = threadIdx.x % LPR;
    const uint32_t row = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    if (row >= rows) return;
    const float *zr = z + (size_t)row * ldz;
    float mx = -INFINITY;
    for (uint32_t c = lane; c < cols; c += LPR) mx = fmaxf(mx, zr[c]);
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (uint32_t c = lane; c < cols; c += LPR) sum += expf(zr[c] - mx);
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    sum += 1e-20f;
    const float *lr = lab + (size_t)row * ldl;
    float *dr = d + (size_t)row * ldd;
    for (uint32_t c = lane; c < cols; c += LPR) {
        float p = expf(zr[c] - mx) / sum;
        const float l = lr[c];
        if (sub_only) {  // predictGAT (engine/ops/gat_ops.cpp:246-265): softmax - label
            dr[c] = p - l;
        } else {
            const uint64_t flat = (uint64_t)row * cols + c;
            if (flat >= mask_first && flat - mask_first < mask_count) p = l;
            dr[c] = (p - l) / inv_mode_denom;
        }
    }
}

static void launch_softmax_rows(uint32_t rows, uint32_t cols, const float *z, uint32_t ldz, const float *lab,
                                uint32_t ldl, float *d, uint32_t ldd, float denom, uint64_t mask_first,
                                uint64_t mask_count, int sub_only, hipStream_t s) {
#define SOFTMAX_LPR(L)                                                                                              \
    hipLaunchKernelGGL(softmax_xent_kernel<L>, dim3((rows + 256 / L - 1) / (256 / L)), dim3(256), 0, s, rows, cols, z, \
                       ldz, lab, ldl, d, ldd, denom, mask_first, mask_count, sub_only)
    if (cols <= 8) SOFTMAX_LPR(8);
    else if (cols <= 16) SOFTMAX_LPR(16);
    else if (cols <= 32) SOFTMAX_LPR(32);
    else if (cols <= 48) SOFTMAX_LPR(16);   // e.g. Reddit's 41 classes: three columns per lane, four rows per wave (81 -> 49 us at
                                            // 232 965 rows against one row per wave with 23 idle lanes and six-step trees)
    else SOFTMAX_LPR(64);
#undef SOFTMAX_LPR
}

// CPUComm::getTrainStat (CPU_comm.cpp:448-462): over validation rows
// [val_stt, val_end): acc += lab[argmax(pred)], loss -= log(pred[argmax(lab)]).
// One row per thread with the reference's own sequential arithmetic, but the rows of a block are staged through
// LDS with coalesced loads first (a thread reading its row straight from HBM touches 64 lines per instruction);
// per-block tree sums, then one block adds the block partials in a fixed order -> deterministic.  (thrust functors
// in the CUDA backend: comp_unit.cu:258-312, cuda_ops.cuh:45-113.)
constexpr int STAT_ROWS_MAX = 64;   // rows per block (fewer when the label set is so wide that 64 rows do not fit LDS)
static uint32_t stat_rows_per_block(uint32_t cols) {
    uint32_t rb = STAT_ROWS_MAX;
    while (rb > 1 && (size_t)2 * rb * (cols + 1) * sizeof(float) > 48 * 1024) rb >>= 1;
    return rb;
}
__global__ __launch_bounds__(256) void train_stat_kernel(uint32_t cols, const float *z, uint32_t ldz,
                                                         const float *lab, uint32_t ldl,
                                                         uint32_t val_stt, uint32_t val_end,
                                                         float *partial, uint32_t STAT_ROWS) {
    extern __shared__ float sm[];            // [STAT_ROWS][cols+1] z, then the same for lab
    const uint32_t pitch = cols + 1;
    float *sz = sm, *sl = sm + STAT_ROWS * pitch;
    const uint32_t r0 = val_stt + blockIdx.x * STAT_ROWS;
    for (uint32_t i = threadIdx.x; i < STAT_ROWS * cols; i += 256) {
        const uint32_t rr = i / cols, c = i % cols;
        const uint32_t r = r0 + rr;
        sz[rr * pitch + c] = r < val_end ? z[(size_t)r * ldz + c] : 0.f;
        sl[rr * pitch + c] = r < val_end ? lab[(size_t)r * ldl + c] : 0.f;
    }
    __syncthreads();
    __shared__ float sa[STAT_ROWS_MAX], sls[STAT_ROWS_MAX];
    if (threadIdx.x < STAT_ROWS) {
        float acc = 0.f, loss = 0.f;
        if (r0 + threadIdx.x < val_end) {
            const float *zr = sz + threadIdx.x * pitch;
            const float *lr = sl + threadIdx.x * pitch;
            float mx = zr[0];
            uint32_t am = 0, al = 0;
            float lmax = lr[0];
            for (uint32_t c = 1; c < cols; ++c) {
                if (zr[c] > mx) { mx = zr[c]; am = c; }      // argmax(pred) == argmax(z), first max
                if (lr[c] > lmax) { lmax = lr[c]; al = c; }
            }
            float den = 1e-20f;
            for (uint32_t c = 0; c < cols; ++c) den += expf(zr[c] - mx);
            acc = lr[am];
            loss = -logf(expf(zr[al] - mx) / den);
        }
        sa[threadIdx.x] = acc;
        sls[threadIdx.x] = loss;
    }
    __syncthreads();
    for (int o = (int)STAT_ROWS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sa[threadIdx.x] += sa[threadIdx.x + o];
            sls[threadIdx.x] += sls[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = sa[0];
        partial[2 * blockIdx.x + 1] = sls[0];
    }
}
// one block: thread t adds partials t, t+256, ... in order, then a fixed tree over the 256 threads
__global__ __launch_bounds__(256) void train_stat_final_kernel(const float *partial, uint32_t nb, float *stat) {
    __shared__ float sa[256], sl[256];
    float a = 0.f, l = 0.f;
    for (uint32_t b = threadIdx.x; b < nb; b += 256) {
        a += partial[2 * b];
        l += partial[2 * b + 1];
    }
    sa[threadIdx.x] = a;
    sl[threadIdx.x] = l;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sa[threadIdx.x] += sa[threadIdx.x + o];
            sl[threadIdx.x] += sl[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stat[0] = sa[0];
        stat[1] = sl[0];
    }
}

size_t softmax_xent_scratch_bytes(uint32_t cols, uint32_t val_rows) {   // per-block partials of the validation statistics
    const uint32_t rb = stat_rows_per_block(cols);
    return (size_t)(2 * ((val_rows + rb - 1) / rb) + 64) * sizeof(float);
}

hipError_t launch_softmax_xent(uint32_t rows, uint32_t cols, const float *z, uint32_t ldz,
                               const float *lab, uint32_t ldl, float *d, uint32_t ldd, float denom,
                               uint32_t val_stt, uint32_t val_end, uint64_t mask_first,
                               uint64_t mask_count, float *stat, float *stat_partial, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    const uint32_t rb = stat_rows_per_block(cols);
    const uint32_t nb = (val_end - val_stt + rb - 1) / rb;
    if (nb)
        hipLaunchKernelGGL(train_stat_kernel, dim3(nb), dim3(256), (size_t)2 * rb * (cols + 1) * sizeof(float), s, cols, z,
                           ldz, lab, ldl, val_stt, val_end, stat_partial, rb);
    hipLaunchKernelGGL(train_stat_final_kernel, dim3(1), dim3(256), 0, s, stat_partial, nb, stat);
    launch_softmax_rows(rows, cols, z, ldz, lab, ldl, d, ldd, denom, mask_first, mask_count, 0, s);
    return hipGetLastError();
}

hipError_t launch_softmax_sub(uint32_t rows, uint32_t cols, const float *z, uint32_t ldz,
                              const float *lab, uint32_t ldl, float *out, uint32_t ldo, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    launch_softmax_rows(rows, cols, z, ldz, lab, ldl, out, ldo, 1.f, 0, 0, 1, s);
    return hipGetLastError();
}

// ---- synthetic inputs -----------------------------------------------------------
// counter RNG: splitmix64(seed ^ (global_row * cols + col)) -> U[lo, hi); the same
// value for a vertex whichever partition holds it (row ids are global ids).
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void fill_uniform_kernel(float *d, uint64_t rows, uint32_t cols, uint32_t ld,
                                    const uint32_t *row_ids, uint64_t seed, float lo, float hi) {
    const uint64_t n = rows * ld;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / ld;
        const uint32_t c = (uint32_t)(i % ld);
        float v = 0.f;
        if (c < cols) {
            const uint64_t gr = row_ids ? row_ids[r] : r;
            const uint64_t h = splitmix64(seed ^ (gr * cols + c));
            const float u = (float)(h >> 40) * (1.0f / 16777216.0f);  // 24 bits -> [0,1)
            v = lo + (hi - lo) * u;
        }
        d[i] = v;
    }
}

hipError_t launch_fill_uniform(float *d, uint64_t rows, uint32_t cols, uint32_t ld, uint64_t seed,
                               float lo, float hi, hipStream_t s) {
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, s, d, rows, cols, ld,
                       (const uint32_t *)nullptr, seed, lo, hi);
    return hipGetLastError();
}

hipError_t launch_fill_uniform_ids(float *d, uint64_t rows, uint32_t cols, uint32_t ld,
                                   const uint32_t *row_ids, uint64_t seed, float lo, float hi,
                                   hipStream_t s) {
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(2048), dim3(256), 0, s, d, rows, cols, ld, row_ids,
                       seed, lo, hi);
    return hipGetLastError();
}

// one-hot expansion of u32 labels (Engine::readLabelsFile, engine/utils.cpp:559-596)
__global__ void onehot_kernel(float *d, uint64_t rows, uint32_t cols, uint32_t ld, const uint32_t *labels) {
    const uint64_t n = rows * ld;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / ld;
        const uint32_t c = (uint32_t)(i % ld);
        d[i] = (c < cols && labels[r] == c) ? 1.f : 0.f;
    }
}

hipError_t launch_onehot(float *d, uint64_t rows, uint32_t cols, uint32_t ld, const uint32_t *labels,
                         hipStream_t s) {
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(onehot_kernel, dim3(1024), dim3(256), 0, s, d, rows, cols, ld, labels);
    return hipGetLastError();
}

// dense (lds == cols) <-> padded (ldd) row copies; destination padding is zeroed
__global__ void pad_copy_kernel(float *dst, uint32_t ldd, const float *src, uint32_t lds, uint64_t rows,
                                uint32_t cols) {
    const uint64_t n = rows * ldd;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / ldd;
        const uint32_t c = (uint32_t)(i % ldd);
        dst[i] = c < cols ? src[r * lds + c] : 0.f;
    }
}

// out[v,:] = base[v,:] + rs[v] * S[v,:]   (base = out itself when accumulating, x otherwise): what is left of a GAT-prototype
// aggregation once the unweighted neighbour sum S is known (abi_stages.hip: the reference's edge scores depend on the destination only)
__global__ __launch_bounds__(256) void row_axpy_kernel(float *out, const float *S, const float *rs, const float *x, uint32_t ld4, uint64_t n4) {
    float4 *o4 = reinterpret_cast<float4 *>(out);
    const float4 *s4 = reinterpret_cast<const float4 *>(S), *x4 = reinterpret_cast<const float4 *>(x ? x : out);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) {
        const float r = rs[i / ld4];
        const float4 b = x4[i], sv = s4[i];
        o4[i] = make_float4(fmaf(r, sv.x, b.x), fmaf(r, sv.y, b.y), fmaf(r, sv.z, b.z), fmaf(r, sv.w, b.w));
    }
}
hipError_t launch_row_axpy(float *out, const float *S, const float *rs, const float *x /* nullptr: out += */, uint64_t rows, uint32_t ld,
                           hipStream_t s) {
    if (rows == 0 || ld == 0) return hipSuccess;
    if (ld & 3) return hipErrorInvalidValue;
    const uint64_t n4 = rows * (ld >> 2);
    hipLaunchKernelGGL(row_axpy_kernel, dim3((uint32_t)std::min<uint64_t>(8192, (n4 + 255) / 256)), dim3(256), 0, s, out, S, rs, x, ld >> 2, n4);
    return hipGetLastError();
}

hipError_t launch_pad_copy(float *dst, uint32_t ldd, const float *src, uint32_t lds, uint64_t rows,
                           uint32_t cols, hipStream_t s) {
    if (rows == 0 || ldd == 0) return hipSuccess;
    hipLaunchKernelGGL(pad_copy_kernel, dim3(2048), dim3(256), 0, s, dst, ldd, src, lds, rows, cols);
    return hipGetLastError();
}

// ---- K5: GAT edge kernels -----------------------------------------------------------
// edgNNForwardGAT (CPU_comm.cpp:190-203) = expandDot (299-319) + leakyRelu (384-395):
//   az[e] = z[dst(e),:] . a ;  A[e] = az > 0 ? az : 0.01 az,  dst(e) = owning column.
// All edges of a column share the value, so one wave computes the row dot once
// and streams it over the column's edge range (no csrRowInd gather as in
// GPU-Computation/comp_server.cu:255-262, comp_unit.cu:93-134).
__global__ __launch_bounds__(256) void edge_forward_gat_kernel(uint32_t N, uint32_t F,
                                                               const uint64_t *colptr, const float *z,
                                                               uint32_t ldz, const float *a, float *az,
                                                               float *A, float *arow, float *azrow) {
    const int lane = threadIdx.x & 63;
    const uint32_t v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= N) return;
    const float *zr = z + (size_t)v * ldz;
    float s = 0.f;
    for (uint32_t j = lane; j < F; j += 64) s = fmaf(zr[j], a[j], s);
    s = wave_sum(s);
    const float act = s > 0.f ? s : 0.01f * s;
    if (az)      // (nullptr: the per-edge copies are written on demand, expand_rows_to_edges)
        for (uint64_t e = colptr[v] + lane; e < colptr[v + 1]; e += 64) {
            az[e] = s;
            A[e] = act;
        }
    if (lane == 0) {
        arow[v] = act;   // every edge of column v carries the same weight
        if (azrow) azrow[v] = s;
    }
}

// per-edge copy of a per-destination value: out[e] = row[dst(e)] for every in-edge of every local vertex
__global__ __launch_bounds__(256) void expand_rows_to_edges_kernel(uint32_t N, const uint64_t *colptr, const float *row, float *out) {
    const int lane = threadIdx.x & 63;
    const uint32_t v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= N) return;
    const float x = row[v];
    for (uint64_t e = colptr[v] + lane; e < colptr[v + 1]; e += 64) out[e] = x;
}
hipError_t launch_expand_rows_to_edges(uint32_t N, const uint64_t *colptr, const float *row, float *out, hipStream_t s) {
    if (N == 0) return hipSuccess;
    hipLaunchKernelGGL(expand_rows_to_edges_kernel, dim3((N + 3) / 4), dim3(256), 0, s, N, colptr, row, out);
    return hipGetLastError();
}

hipError_t launch_edge_forward_gat(uint32_t N, uint32_t F, const uint64_t *colptr, const float *z,
                                   uint32_t ldz, const float *a, float *az, float *A, float *arow,
                                   hipStream_t s, float *azrow) {
    if (N == 0) return hipSuccess;
    hipLaunchKernelGGL(edge_forward_gat_kernel, dim3((N + 3) / 4), dim3(256), 0, s, N, F, colptr, z, ldz, a, az, A, arow, azrow);
    return hipGetLastError();
}

// edgNNBackwardGAT (CPU_comm.cpp:205-242):
//   dLRelu[e] = az[e] > 0 ? 1 : .01 ;  dAct[e,:] = grad[dst(e),:] * dLRelu[e]
//   dA[e] = dAct[e,:] . a ;  r[j] = sum_e dAct[e,j]     (the E x F dAct is never built)
// Per column v: s_v = dLRelu of its edges, t_v = s_v * (grad[v,:] . a);
// dA[e] = t_v; cw[v] = deg(v) * s_v is the column weight for r = grad^T cw.
__global__ __launch_bounds__(256) void edge_backward_gat_kernel(uint32_t N, uint32_t F,
                                                                const uint64_t *colptr,
                                                                const float *grad, uint32_t ldg,
                                                                const float *az, const float *a,
                                                                float *dA, float *cw, float *drow, const float *azrow) {
    const int lane = threadIdx.x & 63;
    const uint32_t v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= N) return;
    const uint64_t e0 = colptr[v], e1 = colptr[v + 1];
    float sv = 0.f;
    if (e1 > e0) sv = (azrow ? azrow[v] : az[e0]) > 0.f ? 1.f : 0.01f;
    const float *gr = grad + (size_t)v * ldg;
    float s = 0.f;
    for (uint32_t j = lane; j < F; j += 64) s = fmaf(gr[j] * sv, a[j], s);
    s = wave_sum(s);
    if (dA)
        for (uint64_t e = e0 + lane; e < e1; e += 64) dA[e] = s;
    if (lane == 0) {
        cw[v] = (float)(e1 - e0) * sv;
        drow[v] = s;
    }
}

hipError_t launch_edge_backward_gat(uint32_t N, uint32_t F, const uint64_t *colptr, const float *grad,
                                    uint32_t ldg, const float *az, const float *a, float *dA, float *cw,
                                    float *drow, hipStream_t s, const float *azrow) {
    if (N == 0) return hipSuccess;
    hipLaunchKernelGGL(edge_backward_gat_kernel, dim3((N + 3) / 4), dim3(256), 0, s, N, F, colptr, grad,
                       ldg, az, a, dA, cw, drow, azrow);
    return hipGetLastError();
}

// y[v] = X[v,:] . r   (wave per row)
__global__ __launch_bounds__(256) void rowdot_kernel(uint32_t N, uint32_t F, const float *X, uint32_t ld,
                                                     const float *r, float *y) {
    const int lane = threadIdx.x & 63;
    const uint32_t v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= N) return;
    const float *xr = X + (size_t)v * ld;
    float s = 0.f;
    for (uint32_t j = lane; j < F; j += 64) s = fmaf(xr[j], r[j], s);
    s = wave_sum(s);
    if (lane == 0) y[v] = s;
}

// partial[b][j] = sum_{v in block b's rows} w[v] * X[v,j];  then out[j] = sum_b partial[b][j]
// out[j] = sum_v w[v] * X[v, j]: column sums weighted per row (the a_i gradient of the GAT prototype, CPU_comm.cpp:232-236).
// Round 6: float4 columns x row groups per workgroup (all 256 threads load, rows in flight in parallel) and a tree-free, fixed-order
// second stage on 32 columns x 8 row groups per workgroup -- 80 + 43 us -> per call at Reddit size before (one thread per column
// walking its block's rows, then one thread per column walking 1 024 partials).  Sums are formed in a fixed order: deterministic.
__global__ __launch_bounds__(256) void colsum_w_kernel(uint32_t N, uint32_t F, const float *X, uint32_t ld,
                                                       const float *w, float *partial, uint32_t rows_per_block) {
    __shared__ float4 red[256];
    const uint32_t r0 = blockIdx.x * rows_per_block;
    const uint32_t r1 = min(N, r0 + rows_per_block);
    const uint32_t F4 = (F + 3) >> 2;                 // float4 columns (rows are padded to 32 floats: the last one reads padding zeros)
    const uint32_t CW = min(F4, 256u);                // columns handled per pass
    const uint32_t RG = 256u / CW;                    // row groups
    const uint32_t c = threadIdx.x % CW, rg = threadIdx.x / CW;
    for (uint32_t c0 = 0; c0 < F4; c0 += CW) {
        const uint32_t col = c0 + c;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < F4 && rg < RG)
            for (uint32_t v = r0 + rg; v < r1; v += RG) {
                const float wv = w[v];
                float4 x;
                if ((ld & 3u) == 0) {
                    x = *reinterpret_cast<const float4 *>(X + (size_t)v * ld + 4 * col);
                } else {                               // (one-column tensors keep ld = 1: element loads, zeros past F)
                    const float *xr = X + (size_t)v * ld + 4 * col;
                    x = make_float4(xr[0], 4 * col + 1 < F ? xr[1] : 0.f, 4 * col + 2 < F ? xr[2] : 0.f, 4 * col + 3 < F ? xr[3] : 0.f);
                }
                s.x = fmaf(wv, x.x, s.x); s.y = fmaf(wv, x.y, s.y); s.z = fmaf(wv, x.z, s.z); s.w = fmaf(wv, x.w, s.w);
            }
        red[threadIdx.x] = s;
        __syncthreads();
        if (rg == 0 && col < F4) {
            for (uint32_t k = 1; k < RG; ++k) {       // row groups in order
                const float4 t = red[k * CW + c];
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            float *p = partial + (size_t)blockIdx.x * F + 4 * col;
            const float r[4] = {s.x, s.y, s.z, s.w};
            for (uint32_t q = 0; q < 4 && 4 * col + q < F; ++q) p[q] = r[q];
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(uint32_t F, const float *partial, uint32_t nb, float *out) {
    __shared__ float red[256];
    const uint32_t c = threadIdx.x & 31u, rg = threadIdx.x >> 5;      // 32 columns x 8 groups of partials
    const uint32_t j = blockIdx.x * 32u + c;
    float s = 0.f;
    if (j < F)
        for (uint32_t b = rg; b < nb; b += 8) s += partial[(size_t)b * F + j];
    red[threadIdx.x] = s;
    __syncthreads();
    if (rg == 0 && j < F) {
        for (uint32_t k = 1; k < 8; ++k) s += red[k * 32 + c];
        out[j] = s;
    }
}

hipError_t launch_rowdot(uint32_t N, uint32_t F, const float *X, uint32_t ld, const float *r, float *y,
                         hipStream_t s) {
    if (N == 0) return hipSuccess;
    hipLaunchKernelGGL(rowdot_kernel, dim3((N + 3) / 4), dim3(256), 0, s, N, F, X, ld, r, y);
    return hipGetLastError();
}

// out[F] = X^T w, X is N x F; partial must hold nb*F floats
hipError_t launch_colsum_w(uint32_t N, uint32_t F, const float *X, uint32_t ld, const float *w,
                           float *partial, size_t partial_bytes, float *out, hipStream_t s) {
    if (F == 0) return hipSuccess;
    uint32_t nb = 512;
    while (nb > 1 && (nb > N / 64 || (size_t)nb * F * sizeof(float) > partial_bytes)) nb >>= 1;   // >= 64 rows per block
    const uint32_t rpb = (N + nb - 1) / nb > 0 ? (N + nb - 1) / nb : 1;
    nb = (N + rpb - 1) / rpb;
    if (nb == 0) nb = 1;
    hipLaunchKernelGGL(colsum_w_kernel, dim3(nb), dim3(256), 0, s, N, F, X, ld, w, partial, rpb);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((F + 31) / 32), dim3(256), 0, s, F, partial, nb, out);
    return hipGetLastError();
}

// ---- K6: halo pack / unpack ------------------------------------------------------------
// pack:   dst[i, 0:cols] (dense) = src[rows[i], 0:cols]   (Engine::verticesPushOut's
//         memcpy loop, engine/utils.cpp:640-648, minus the 4-byte gvid per row)
// unpack: dst[rows[i], 0:cols] = src[i, 0:cols] (dense)   (ghostReceiverGCN's memcpy
//         loop at globalToGhostVtcs[gvid] - localVtxCnt, gcn_ops.cpp:310-318)
// cols is a multiple of 4 floats wherever ld is (pack buffers use the padded width).
__global__ void gather_rows_kernel(float *dst, const float *src, uint32_t ld, uint32_t cols4,
                                   const uint32_t *rows, uint32_t n) {
    const uint64_t total = (uint64_t)n * cols4;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / cols4);
        const uint32_t c = (uint32_t)(i % cols4);
        reinterpret_cast<float4 *>(dst)[i] =
            reinterpret_cast<const float4 *>(src + (size_t)rows[r] * ld)[c];
    }
}
__global__ void scatter_rows_kernel(float *dst, const float *src, uint32_t ld, uint32_t cols4,
                                    const uint32_t *rows, uint32_t n) {
    const uint64_t total = (uint64_t)n * cols4;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / cols4);
        const uint32_t c = (uint32_t)(i % cols4);
        reinterpret_cast<float4 *>(dst + (size_t)rows[r] * ld)[c] =
            reinterpret_cast<const float4 *>(src)[i];
    }
}

hipError_t launch_gather_rows(float *dst, const float *src, uint32_t ld, uint32_t cols,
                              const uint32_t *rows, uint32_t n, hipStream_t s) {
    if (n == 0 || cols == 0) return hipSuccess;
    if ((cols & 3) || (ld & 3)) return hipErrorInvalidValue;   // float4 rows only (every exchanged tensor is padded to 32 floats)
    const uint64_t total = (uint64_t)n * (cols / 4);
    int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, s, dst, src, ld, cols / 4, rows, n);
    return hipGetLastError();
}
hipError_t launch_scatter_rows(float *dst, const float *src, uint32_t ld, uint32_t cols,
                               const uint32_t *rows, uint32_t n, hipStream_t s) {
    if (n == 0 || cols == 0) return hipSuccess;
    if ((cols & 3) || (ld & 3)) return hipErrorInvalidValue;
    const uint64_t total = (uint64_t)n * (cols / 4);
    int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(blocks), dim3(256), 0, s, dst, src, ld, cols / 4, rows, n);
    return hipGetLastError();
}

// ---- K7: Adam -----------------------------------------------------------------------------
// AdamOptimizer::update (reference src/weight-server/AdamOptimizer.cpp:36-51), the
// mixed float/double expressions kept as written there ("(1. - BETA1) * gt" is double).
__global__ void adam_kernel(float *w, const float *g, float *m, float *v, uint64_t n, float lr_t) {
    const float BETA1 = .9f, BETA2 = .999f, EPSILON = 1e-07f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const float gt = g[i];
        const float pm = m[i], pd = v[i];
        const float nm = (float)(BETA1 * pm + (1. - BETA1) * gt);
        const float nv = (float)(BETA2 * pd + (1. - BETA2) * gt * gt);
        m[i] = nm;
        v[i] = nv;
        const float delta = (float)(lr_t * nm / (sqrt((double)nv) + EPSILON));
        w[i] -= delta;
    }
}

hipError_t launch_adam(float *w, const float *g, float *m, float *v, uint64_t n, float lr_t,
                       hipStream_t s) {
    if (n == 0) return hipSuccess;
    int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, w, g, m, v, n, lr_t);
    return hipGetLastError();
}

// the same update inside a replayed epoch graph: the step size of replay number *idx comes
// from a table the host filled before the launches (kernel arguments are frozen at capture)
__global__ void adam_table_kernel(float *w, const float *g, float *m, float *v, uint64_t n, const float *lr_table,
                                  const uint32_t *idx) {
    const float BETA1 = .9f, BETA2 = .999f, EPSILON = 1e-07f;
    const float lr_t = lr_table[*idx];
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        const float gt = g[i];
        const float pm = m[i], pd = v[i];
        const float nm = (float)(BETA1 * pm + (1. - BETA1) * gt);
        const float nv = (float)(BETA2 * pd + (1. - BETA2) * gt * gt);
        m[i] = nm;
        v[i] = nv;
        const float delta = (float)(lr_t * nm / (sqrt((double)nv) + EPSILON));
        w[i] -= delta;
    }
}
__global__ void bump_counter_kernel(uint32_t *idx) { *idx += 1u; }

hipError_t launch_adam_table(float *w, const float *g, float *m, float *v, uint64_t n, const float *lr_table,
                             const uint32_t *idx, hipStream_t s) {
    if (n == 0) return hipSuccess;
    int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(adam_table_kernel, dim3(blocks), dim3(256), 0, s, w, g, m, v, n, lr_table, idx);
    return hipGetLastError();
}
hipError_t launch_bump_counter(uint32_t *idx, hipStream_t s) {
    hipLaunchKernelGGL(bump_counter_kernel, dim3(1), dim3(1), 0, s, idx);
    return hipGetLastError();
}

}  // namespace dory
