// gather_probe.hip -- what rate can a row gather reach when every source row is L2-resident?  Each lane group of LPR
// lanes reads U random rows (LPR*16 B each) of a per-XCD window per iteration; indices come from an LCG so no index
// stream competes.  MODE 0: global_load_dwordx4 into registers.  MODE 1: global_load_lds_dwordx4 (LDS-DMA) into a
// per-wave staging area, read back with ds_read_b128.  MODE 2: streaming (consecutive rows) for the L2 ceiling.
//   hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip && ./gather_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);       \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

__device__ __forceinline__ float4 add4(float4 x, float4 a) {
    a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
    return a;
}

template <int LPR, int U, int MODE>
__global__ __launch_bounds__(256) void gather_probe(const float4 *__restrict__ x, uint32_t win_rows, uint32_t stride4,
                                                    int iters, float4 *__restrict__ out) {
    __shared__ float4 stage[MODE == 1 ? 4 * U * 64 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane % LPR;
    const uint32_t gid = (blockIdx.x * 256 + threadIdx.x) / LPR;
    const uint32_t xcd = blockIdx.x & 7u;
    const float4 *xw = x + (size_t)xcd * win_rows * stride4;
    uint32_t state = gid * 2654435761u + 12345u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0 || MODE == 2) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t r;
                if constexpr (MODE == 0) {
                    state = state * 1664525u + 1013904223u;
                    r = (uint32_t)(((uint64_t)(state >> 4) * win_rows) >> 28);
                } else {
                    r = (gid + (uint32_t)(it * U + u) * 1031u) % win_rows;
                }
                v[u] = xw[(size_t)r * stride4 + li];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc = add4(v[u], acc);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                state = state * 1664525u + 1013904223u;
                const uint32_t r = (uint32_t)(((uint64_t)(state >> 4) * win_rows) >> 28);
                const float4 *src = xw + (size_t)r * stride4 + li;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(stage + (wave * U + u) * 64),
                                                 16, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
#pragma unroll
            for (int u = 0; u < U; ++u) acc = add4(stage[(wave * U + u) * 64 + lane], acc);
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int LPR, int U, int MODE>
static void run(const float4 *dx, uint32_t win_rows, uint32_t stride4, float4 *dout, int wg_per_cu) {
    const int nwg = 256 * wg_per_cu, iters = 2000 / U;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((gather_probe<LPR, U, MODE>), dim3(nwg), dim3(256), 0, 0, dx, win_rows, stride4, iters, dout);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)nwg * 256 * iters * U * 16;
    printf("mode=%d lanes/row=%2d (%4d B) U=%2d window=%5.2f MB stride=%4u B wg/cu=%d: %7.3f ms  %6.2f TB/s\n", MODE, LPR,
           LPR * 16, U, (double)win_rows * LPR * 16 / 1048576.0, stride4 * 16, wg_per_cu, ms, bytes / (ms * 1e-3) / 1e12);
    fflush(stdout);
}


// K1s-shaped probe: one 1024-thread workgroup per CU, every lane group reads (row, weight) pairs from its own LDS stage
// (filled once from an LCG), U gathers per batch, the batch summed into one of R accumulators.  SRC 0: addresses from
// the LCG (no LDS in the chain), SRC 1: from LDS.
template <int U, int R, int SRC>
__global__ __launch_bounds__(1024) void k1s_probe(const float4 *__restrict__ x, uint32_t win_rows, uint32_t stride4, int iters,
                                                  float4 *__restrict__ out) {
    __shared__ uint2 stage[32][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, g = wave * 2 + (lane >> 5);
    const uint32_t xcd = blockIdx.x & 7u;
    const float4 *xw = x + (size_t)xcd * win_rows * stride4 + li;
    uint32_t state = (blockIdx.x * 32 + g) * 2654435761u + 12345u;
    for (int i = li; i < 128; i += 32) {
        uint32_t st2 = state + i * 7919u;
        st2 = st2 * 1664525u + 1013904223u;
        stage[g][i] = make_uint2((uint32_t)(((uint64_t)(st2 >> 4) * win_rows) >> 28), 0x3f800000u);
    }
    __syncthreads();
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint2 *st = stage[g];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int e = ((it * R + r) * U) & 127 & ~(U - 1);
            uint2 en[U];
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (SRC == 1) en[u] = st[(e + u) & 127];
                else {
                    state = state * 1664525u + 1013904223u;
                    en[u] = make_uint2((uint32_t)(((uint64_t)(state >> 4) * win_rows) >> 28), 0x3f800000u);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = xw[(size_t)en[u].x * stride4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float w = __uint_as_float(en[u].y);
                acc[r].x += w * v[u].x; acc[r].y += w * v[u].y; acc[r].z += w * v[u].z; acc[r].w += w * v[u].w;
            }
        }
    }
    float4 t = acc[0];
#pragma unroll
    for (int r = 1; r < R; ++r) t = add4(acc[r], t);
    out[(size_t)blockIdx.x * 1024 + threadIdx.x] = t;
}

template <int U, int R, int SRC>
static void run_k1s(const float4 *dx, uint32_t win_rows, uint32_t stride4, float4 *dout) {
    const int nwg = 256, iters = 4000 / (U * R);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k1s_probe<U, R, SRC>), dim3(nwg), dim3(1024), 0, 0, dx, win_rows, stride4, iters, dout);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)nwg * 1024 * iters * U * R * 16;
    printf("k1s-shaped: 1024 threads/CU U=%d R=%2d addresses from %s: %7.3f ms  %6.2f TB/s\n", U, R, SRC ? "LDS" : "LCG", ms,
           bytes / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

// the same with K1s's row loop: every row has n in [LO, LO + SPAN) entries (different in the two lane groups of a wave),
// full batches of U then one predicated tail, as spmm_sweep_kernel does.  Rate counts the entries gathered.
template <int U, int R, int LO, int SPAN>
__global__ __launch_bounds__(1024) void k1s_rows_probe(const float4 *__restrict__ x, uint32_t win_rows, uint32_t stride4, int iters,
                                                       float4 *__restrict__ out, unsigned long long *__restrict__ nent) {
    __shared__ uint2 stage[32][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, g = wave * 2 + (lane >> 5);
    const uint32_t xcd = blockIdx.x & 7u;
    const float4 *xw = x + (size_t)xcd * win_rows * stride4 + li;
    uint32_t state = (blockIdx.x * 32 + g) * 2654435761u + 12345u;
    for (int i = li; i < 128; i += 32) {
        uint32_t st2 = state + i * 7919u;
        st2 = st2 * 1664525u + 1013904223u;
        stage[g][i] = make_uint2((uint32_t)(((uint64_t)(st2 >> 4) * win_rows) >> 28), 0x3f800000u);
    }
    __syncthreads();
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint2 *st = stage[g];
    unsigned long long cnt = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t e0 = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            state = state * 1664525u + 1013904223u;
            const uint32_t n = LO + (state >> 8) % SPAN;
            const uint32_t hi = e0 + n;
            cnt += n;
            uint32_t e = e0;
            for (; e + U <= hi; e += U) {
                uint2 en[U];
                float4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) en[u] = st[(e + u) & 127];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = xw[(size_t)en[u].x * stride4];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float w = __uint_as_float(en[u].y);
                    acc[r].x += w * v[u].x; acc[r].y += w * v[u].y; acc[r].z += w * v[u].z; acc[r].w += w * v[u].w;
                }
            }
            if (e < hi) {
                const uint32_t m = hi - e;
                uint2 en[U - 1];
                float4 v[U - 1];
#pragma unroll
                for (int u = 0; u < U - 1; ++u) en[u] = st[(min(e + u, hi - 1)) & 127];
#pragma unroll
                for (int u = 0; u < U - 1; ++u) v[u] = (uint32_t)u < m ? xw[(size_t)en[u].x * stride4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < U - 1; ++u) {
                    const float w = (uint32_t)u < m ? __uint_as_float(en[u].y) : 0.f;
                    acc[r].x += w * v[u].x; acc[r].y += w * v[u].y; acc[r].z += w * v[u].z; acc[r].w += w * v[u].w;
                }
            }
            e0 = hi & 127;
        }
    }
    float4 t = acc[0];
#pragma unroll
    for (int r = 1; r < R; ++r) t = add4(acc[r], t);
    out[(size_t)blockIdx.x * 1024 + threadIdx.x] = t;
    if (li == 0) atomicAdd(nent, cnt);
}

template <int U, int R, int LO, int SPAN>
static void run_rows(const float4 *dx, uint32_t win_rows, uint32_t stride4, float4 *dout) {
    const int nwg = 256, iters = 4000 / (10 * R);
    unsigned long long *dn, hn = 0;
    CK(hipMalloc(&dn, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(dn, 0, 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k1s_rows_probe<U, R, LO, SPAN>), dim3(nwg), dim3(1024), 0, 0, dx, win_rows, stride4, iters, dout, dn);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&hn, dn, 8, hipMemcpyDeviceToHost));
    const double bytes = (double)hn * 512;
    printf("k1s rows: U=%d R=%2d row length %d..%d: %7.3f ms  %6.2f TB/s\n", U, R, LO, LO + SPAN - 1, ms, bytes / (ms * 1e-3) / 1e12);
    fflush(stdout);
    CK(hipFree(dn));
}

// paired halves: the two 32-lane halves of a wave gather entries e and e+1 of the SAME row (one row at a time per wave,
// R rows per wave), so the halves never run different trip counts; full batches of U instructions (2U entries), one
// predicated tail.  Rate counts the entries gathered.
template <int U, int R, int LO, int SPAN>
__global__ __launch_bounds__(1024) void k1s_pair_probe(const float4 *__restrict__ x, uint32_t win_rows, uint32_t stride4, int iters,
                                                       float4 *__restrict__ out, unsigned long long *__restrict__ nent) {
    __shared__ uint2 stage[16][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const uint32_t xcd = blockIdx.x & 7u;
    const float4 *xw = x + (size_t)xcd * win_rows * stride4 + li;
    uint32_t state = (blockIdx.x * 16 + wave) * 2654435761u + 12345u;
    for (int i = lane; i < 256; i += 64) {
        uint32_t st2 = state + i * 7919u;
        st2 = st2 * 1664525u + 1013904223u;
        stage[wave][i] = make_uint2((uint32_t)(((uint64_t)(st2 >> 4) * win_rows) >> 28), 0x3f800000u);
    }
    __syncthreads();
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint2 *st = stage[wave];
    unsigned long long cnt = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t e0 = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            state = state * 1664525u + 1013904223u;
            const uint32_t n = LO + (state >> 8) % SPAN;
            const uint32_t hi = e0 + n;
            cnt += n;
            uint32_t e = e0;
            for (; e + 2 * U <= hi; e += 2 * U) {
                uint2 en[U];
                float4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) en[u] = st[(e + 2 * u + h) & 255];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = xw[(size_t)en[u].x * stride4];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float w = __uint_as_float(en[u].y);
                    acc[r].x += w * v[u].x; acc[r].y += w * v[u].y; acc[r].z += w * v[u].z; acc[r].w += w * v[u].w;
                }
            }
            if (e < hi) {
                uint2 en[U];
                float4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) en[u] = st[(min(e + 2 * u + h, hi - 1)) & 255];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = e + 2 * u + h < hi ? xw[(size_t)en[u].x * stride4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float w = e + 2 * u + h < hi ? __uint_as_float(en[u].y) : 0.f;
                    acc[r].x += w * v[u].x; acc[r].y += w * v[u].y; acc[r].z += w * v[u].z; acc[r].w += w * v[u].w;
                }
            }
            e0 = hi & 255;
        }
    }
    float4 t = acc[0];
#pragma unroll
    for (int r = 1; r < R; ++r) t = add4(acc[r], t);
    out[(size_t)blockIdx.x * 1024 + threadIdx.x] = t;
    if (lane == 0) atomicAdd(nent, cnt);
}

template <int U, int R, int LO, int SPAN>
static void run_pair(const float4 *dx, uint32_t win_rows, uint32_t stride4, float4 *dout) {
    const int nwg = 256, iters = 4000 / (5 * R);
    unsigned long long *dn, hn = 0;
    CK(hipMalloc(&dn, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(dn, 0, 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k1s_pair_probe<U, R, LO, SPAN>), dim3(nwg), dim3(1024), 0, 0, dx, win_rows, stride4, iters, dout, dn);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&hn, dn, 8, hipMemcpyDeviceToHost));
    const double bytes = (double)hn * 512;
    printf("k1s paired halves: U=%d R=%2d row length %d..%d: %7.3f ms  %6.2f TB/s\n", U, R, LO, LO + SPAN - 1, ms, bytes / (ms * 1e-3) / 1e12);
    fflush(stdout);
    CK(hipFree(dn));
}

// wave stream ("K1w"): the R rows of a WAVE are one stream of entries; instruction i carries entries 2i (lanes 0-31) and
// 2i+1 (lanes 32-63); batches of U instructions are always full and straddle row boundaries; a batch is applied to every
// row it overlaps (static accumulator per row, unrolled) with per-lane masks by entry index.  No per-row tails, no
// divergence between the halves.  Row offsets come from LDS as in spmm_sweep_kernel.  Rate counts the entries gathered.
template <int U, int R, int LO, int SPAN>
__global__ __launch_bounds__(1024) void k1w_probe(const float4 *__restrict__ x, uint32_t win_rows, uint32_t stride4, int iters,
                                                  float4 *__restrict__ out, unsigned long long *__restrict__ nent) {
    __shared__ uint2 stage[16][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t row_b = stride4 * 16u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float4 *>(x + (size_t)xcd * win_rows * stride4), 0, win_rows * row_b, 0x00020000);
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    uint32_t state = (blockIdx.x * 16 + wave) * 2654435761u + 12345u;
    for (int i = lane; i < 512; i += 64) {
        uint32_t st2 = state + i * 7919u;
        st2 = st2 * 1664525u + 1013904223u;
        stage[wave][i] = make_uint2((uint32_t)(((uint64_t)(st2 >> 4) * win_rows) >> 28), 0x3f800000u);
    }
    __syncthreads();
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint2 *st = stage[wave];
    uint32_t cnt = 0;
    for (int it = 0; it < iters; ++it) {
        // row offsets of this step: lane l holds the offset of row l (exclusive scan of the lengths), lane R the total
        uint32_t my_off;
        {
            uint32_t s2 = (state + (uint32_t)it * 977u + (uint32_t)lane * 7919u) * 1664525u + 1013904223u;
            uint32_t n = lane < R ? LO + (s2 >> 8) % SPAN : 0u;
            uint32_t pre = n;
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)pre, d, 64);
                if (lane >= d) pre += t;
            }
            my_off = pre - n;
        }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)my_off, R);
        cnt += total;
        uint32_t bpos = 0;
        float w[U];
        u4 v[U];
        bool need = true;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)my_off, r);
            const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)my_off, r + 1);
            if (s == e) continue;
            do {
                if (need) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint2 en = st[(bpos + 2 * u + h) & 511];
                        w[u] = __uint_as_float(en.y);
                        uint32_t off = __umul24(en.x, row_b) + li * 16u;
                        off = bpos + 2 * u + h < total ? off : 0xFFFFFFFFu;
                        v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
                    }
                    need = false;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t ei = bpos + 2 * u + h;
                    const float wm = (ei - s < e - s) ? w[u] : 0.f;
                    acc[r].x += wm * __uint_as_float(v[u].x); acc[r].y += wm * __uint_as_float(v[u].y);
                    acc[r].z += wm * __uint_as_float(v[u].z); acc[r].w += wm * __uint_as_float(v[u].w);
                }
                if (bpos + 2 * U > e) break;          // the row ends inside this batch: the next row shares it
                bpos += 2 * U;                        // batch used up
                need = true;
            } while (bpos < e);
        }
    }
    float4 t = acc[0];
#pragma unroll
    for (int r = 1; r < R; ++r) t = add4(acc[r], t);
    out[(size_t)blockIdx.x * 1024 + threadIdx.x] = t;
    if (lane == 0) atomicAdd(nent, (unsigned long long)cnt);
}

template <int U, int R, int LO, int SPAN>
static void run_k1w(const float4 *dx, uint32_t win_rows, uint32_t stride4, float4 *dout) {
    const int nwg = 256, iters = 8000 / (10 * R);
    unsigned long long *dn, hn = 0;
    CK(hipMalloc(&dn, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(dn, 0, 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k1w_probe<U, R, LO, SPAN>), dim3(nwg), dim3(1024), 0, 0, dx, win_rows, stride4, iters, dout, dn);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&hn, dn, 8, hipMemcpyDeviceToHost));
    const double bytes = (double)hn * 512;
    printf("k1w wave stream: U=%d R=%2d row length %d..%d: %7.3f ms  %6.2f TB/s\n", U, R, LO, LO + SPAN - 1, ms, bytes / (ms * 1e-3) / 1e12);
    fflush(stdout);
    CK(hipFree(dn));
}

int main(int argc, char **argv) {
    const bool only_new = argc > 1;
    const uint32_t stride4 = 152;                 // 608 floats per row, as x at F=602
    const size_t rows_total = 8u * 65536u;
    float4 *dx, *dout;
    CK(hipMalloc(&dx, rows_total * stride4 * 16));
    CK(hipMemset(dx, 0, rows_total * stride4 * 16));
    CK(hipMalloc(&dout, (size_t)256 * 8 * 256 * 16));
    if (only_new) {
        printf("-- references: fixed full batches, K1s row loop, paired halves\n");
        run_k1s<4, 10, 1>(dx, 4096, stride4, dout);
        run_rows<4, 10, 4, 13>(dx, 4096, stride4, dout);
        run_pair<4, 10, 4, 13>(dx, 4096, stride4, dout);
        printf("-- wave stream\n");
        run_k1w<4, 16, 4, 13>(dx, 4096, stride4, dout);
        run_k1w<4, 20, 4, 13>(dx, 4096, stride4, dout);
        run_k1w<4, 10, 4, 13>(dx, 4096, stride4, dout);
        run_k1w<2, 16, 4, 13>(dx, 4096, stride4, dout);
        run_k1w<3, 16, 4, 13>(dx, 4096, stride4, dout);
        run_k1w<6, 16, 4, 13>(dx, 4096, stride4, dout);
        run_k1w<4, 16, 1, 19>(dx, 4096, stride4, dout);
        run_k1w<4, 16, 2, 7>(dx, 4096, stride4, dout);
        run_k1w<4, 16, 8, 25>(dx, 4096, stride4, dout);
        return 0;
    }
    // window sizes in rows chosen for ~2 MB of touched bytes
    printf("-- registers, 2 MB window, lanes/row x in-flight\n");
    run<8, 4, 0>(dx, 16384, stride4, dout, 8);
    run<8, 8, 0>(dx, 16384, stride4, dout, 8);
    run<16, 4, 0>(dx, 8192, stride4, dout, 8);
    run<16, 8, 0>(dx, 8192, stride4, dout, 8);
    run<32, 4, 0>(dx, 4096, stride4, dout, 8);
    run<32, 8, 0>(dx, 4096, stride4, dout, 8);
    run<32, 16, 0>(dx, 4096, stride4, dout, 8);
    run<64, 4, 0>(dx, 2048, stride4, dout, 8);
    run<64, 8, 0>(dx, 2048, stride4, dout, 8);
    printf("-- window size (512-B rows, U=8)\n");
    run<32, 8, 0>(dx, 1024, stride4, dout, 8);
    run<32, 8, 0>(dx, 2048, stride4, dout, 8);
    run<32, 8, 0>(dx, 6144, stride4, dout, 8);
    run<32, 8, 0>(dx, 8192, stride4, dout, 8);
    run<32, 8, 0>(dx, 10240, stride4, dout, 8);
    run<32, 8, 0>(dx, 16384, stride4, dout, 8);
    printf("-- occupancy (512-B rows, 2 MB)\n");
    run<32, 8, 0>(dx, 4096, stride4, dout, 2);
    run<32, 8, 0>(dx, 4096, stride4, dout, 4);
    printf("-- dense rows (stride = row bytes)\n");
    run<32, 8, 0>(dx, 4096, 32, dout, 8);
    run<16, 8, 0>(dx, 8192, 16, dout, 8);
    printf("-- one 1024-thread workgroup per CU (16 waves), 512-B rows, 2 MB window\n");
    run_k1s<4, 1, 0>(dx, 4096, stride4, dout);
    run_k1s<4, 10, 0>(dx, 4096, stride4, dout);
    run_k1s<4, 1, 1>(dx, 4096, stride4, dout);
    run_k1s<4, 10, 1>(dx, 4096, stride4, dout);
    run_k1s<8, 1, 1>(dx, 4096, stride4, dout);
    run_k1s<8, 6, 1>(dx, 4096, stride4, dout);
    run_k1s<2, 10, 1>(dx, 4096, stride4, dout);
    printf("-- K1s row loop (full batches + predicated tail), row lengths differ between the two lane groups of a wave\n");
    run_rows<4, 10, 12, 1>(dx, 4096, stride4, dout);
    run_rows<4, 10, 10, 1>(dx, 4096, stride4, dout);
    run_rows<4, 10, 4, 13>(dx, 4096, stride4, dout);
    run_rows<4, 10, 1, 19>(dx, 4096, stride4, dout);
    run_rows<2, 10, 4, 13>(dx, 4096, stride4, dout);
    run_rows<8, 10, 4, 13>(dx, 4096, stride4, dout);
    run_pair<4, 10, 4, 13>(dx, 4096, stride4, dout);
    run_pair<4, 16, 4, 13>(dx, 4096, stride4, dout);
    run_pair<4, 10, 1, 19>(dx, 4096, stride4, dout);
    run_pair<2, 16, 4, 13>(dx, 4096, stride4, dout);
    run_pair<3, 16, 4, 13>(dx, 4096, stride4, dout);
    run_pair<4, 16, 8, 25>(dx, 4096, stride4, dout);
    run_rows<4, 10, 8, 25>(dx, 4096, stride4, dout);
    printf("-- LDS-DMA gather\n");
    run<16, 4, 1>(dx, 8192, stride4, dout, 8);
    run<32, 4, 1>(dx, 4096, stride4, dout, 8);
    run<32, 8, 1>(dx, 4096, stride4, dout, 8);
    run<64, 8, 1>(dx, 2048, stride4, dout, 8);
    printf("-- streaming from the window (L2 ceiling)\n");
    run<64, 8, 2>(dx, 2048, 64, dout, 8);
    run<32, 8, 2>(dx, 4096, 32, dout, 8);
    return 0;
}
