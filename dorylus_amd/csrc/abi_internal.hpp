// abi_internal.hpp -- what the translation units of the C-ABI share (abi_context.hip: context, tensor table,
// uploads, options; abi_stages.hip: aggregate / apply_vertex / apply_edge; abi_comm.hip: halo exchange, weight
// update, epoch graph).  Not part of the public interface (include/dorylus_hip.h is).
#ifndef DORY_ABI_INTERNAL_HPP
#define DORY_ABI_INTERNAL_HPP
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <random>

#include "ctx.hpp"

namespace dory {

// records the message for dory_last_error and returns `code`
int fail(dory_ctx *c, int code, const char *fmt, ...);

#define HIPCK(c, call)                                                                 \
    do {                                                                               \
        hipError_t e__ = (call);                                                       \
        if (e__ != hipSuccess)                                                         \
            return fail((c), DORY_ERR_HIP, "%s failed: %s (%s:%d)", #call,             \
                        hipGetErrorString(e__), __FILE__, __LINE__);                   \
    } while (0)
#define NCCLCK(c, call)                                                                \
    do {                                                                               \
        ncclResult_t r__ = (call);                                                     \
        if (r__ != ncclSuccess)                                                        \
            return fail((c), DORY_ERR_COMM, "%s failed: %s (%s:%d)", #call,            \
                        ncclGetErrorString(r__), __FILE__, __LINE__);                  \
    } while (0)
#define CHECK_CTX(c)                                                                   \
    if (!(c)) return DORY_ERR_ARG;                                                     \
    std::lock_guard<std::mutex> lock__((c)->mu);                                       \
    HIPCK((c), hipSetDevice((c)->device))

#define NEED(ptr, l, nm)                                                                  \
    Tensor *ptr = find(c, (l), nm);                                                       \
    if (!ptr) return fail(c, DORY_ERR_ARG, "%s: tensor '%s'@%u missing", __func__, nm, (unsigned)(l))

// ---- timing: HIP events on the stream the kernels run on --------------------------
struct Timed {
    dory_ctx *c;
    hipStream_t s;
    const char *fam;
    hipEvent_t a = nullptr, b = nullptr;
    Timed(dory_ctx *ctx, const char *family, hipStream_t st) : c(ctx), s(st), fam(family) {
        if (!c->timing || c->capturing) return;
        if (c->ev_pool.empty()) {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
        } else {
            a = c->ev_pool.back().first;
            b = c->ev_pool.back().second;
            c->ev_pool.pop_back();
        }
        (void)hipEventRecord(a, s);
    }
    ~Timed() {
        if (!c->timing || c->capturing) return;
        (void)hipEventRecord(b, s);
        c->pending.push_back({fam, a, b});
    }
};
// Overlap bookkeeping (timing on only): the exchange whose ghosts are still landing when the compute stream goes on is
// recorded as family "halo_deferred", the launch that runs beside it (local-source blocks / interior rows) as
// "spmm_beside_halo"; drain_timing intersects the two intervals on the device's clock into "halo_hidden".
// bench.py: multi_gpu.halo_overlap_fraction = halo_hidden / halo_deferred.

void drain_timing(dory_ctx *c);
int alloc_tensor(dory_ctx *c, Tensor &t, uint64_t rows, uint32_t cols);
Tensor *find(dory_ctx *c, uint32_t layer, const char *name);
Tensor *findw(std::vector<std::map<std::string, Tensor>> &tab, uint32_t layer, const char *name);
void free_table(std::vector<std::map<std::string, Tensor>> &tab);
int ensure_scratch(dory_ctx *c, size_t bytes);
int ensure_sweep(dory_ctx *c, bool csc, int group);
// drops a recorded epoch (hipGraph): anything that frees or moves what the recorded kernels point at calls this
void epoch_graph_drop_locked(dory_ctx *c);
std::vector<uint32_t> degree_order(const uint64_t *ptr, uint32_t N);
int gemm(dory_ctx *c, int ta, int tb, uint32_t M, uint32_t N, uint32_t K, const Tensor &A, const Tensor &B, Tensor &C,
         Tensor *C2 = nullptr);

template <typename T>
int upload_array(dory_ctx *c, T **dst, const T *src, uint64_t n) {
    size_t b = n * sizeof(T);
    HIPCK(c, hipMalloc((void **)dst, b ? b : 256));
    if (b) HIPCK(c, hipMemcpy(*dst, src, b, hipMemcpyHostToDevice));
    return DORY_OK;
}

// Ghost rows of the last halo exchange land on the comm stream; with "halo_overlap" the compute stream is only
// made to wait for them (event ev_b) by the first consumer.
int wait_halo(dory_ctx *c);
// in-process device transport: second half of a deferred exchange (abi_comm.hip); wait_halo and dory_sync run it
int local_exchange_finish(dory_ctx *c);
// GAT prototype, lazy per-edge tensors (ctx.hpp): bring az@layer (1), "A" (2), dA@layer (4) up to date before somebody reads them
int gat_materialize(dory_ctx *c, uint32_t layer, int which);
// transform-first order applies to this GCN layer (option, model shape, adjacency values): see abi_context.hip
bool tf_layer(dory_ctx *c, uint32_t layer);
bool tf_active(dory_ctx *c);   // = tf_layer(c, 0)
// one all-to-all-v of rows with the plan of `dir` (abi_comm.hip)
int exchange_rows(dory_ctx *c, int dir, Tensor *src, Tensor *ghost, bool defer);
// K1b bookkeeping (abi_stages.hip)
int ensure_blocked(dory_ctx *c, bool csc, int group, bool narrow_set = false /* the multi-head GAT contexts' second pair (ctx.hpp) */);
// the blocked copy a multi-head GAT layer of leading dimension ld gathers through
BlockedAdj &gatmh_blocked_for(dory_ctx *c, bool csc, uint32_t ld);
int blk_group_for(dory_ctx *c, uint32_t ld);

}  // namespace dory
#endif
