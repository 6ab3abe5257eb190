// abi.hip -- host side of the C-ABI (include/dorylus_hip.h): context, device
// tensor table, graph upload and the stage dispatch that replaces the reference's
// Engine::aggregate* / ResourceComm::NNCompute / Engine::scatter* bodies.
// Reference paths are relative to src/graph-server/ unless they start with src/.
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <random>

#include "ctx.hpp"

using namespace dory;

static std::string g_create_err;

static int fail(dory_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}

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

// ---- timing: HIP events on the stream the kernels run on --------------------------
namespace {
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

void drain_timing(dory_ctx *c) {
    for (auto &p : c->pending) {
        (void)hipEventSynchronize(p.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            c->times[p.fam].total_ms += ms;
            c->times[p.fam].launches += 1;
        }
        c->ev_pool.push_back({p.a, p.b});
    }
    c->pending.clear();
}

int alloc_tensor(dory_ctx *c, Tensor &t, uint64_t rows, uint32_t cols) {
    t.rows = rows;
    t.cols = cols;
    t.ld = pad_ld(cols);
    t.owned = true;
    t.d = nullptr;
    size_t b = t.bytes();
    if (b == 0) b = 256;  // keep a valid pointer for empty ghosts
    HIPCK(c, hipMalloc((void **)&t.d, b));
    HIPCK(c, hipMemsetAsync(t.d, 0, b, c->compute));
    return DORY_OK;
}

Tensor *find(dory_ctx *c, uint32_t layer, const char *name) {
    if (layer >= c->tensors.size()) return nullptr;
    auto it = c->tensors[layer].find(name);
    return it == c->tensors[layer].end() ? nullptr : &it->second;
}
Tensor *findw(std::vector<std::map<std::string, Tensor>> &tab, uint32_t layer, const char *name) {
    if (layer >= tab.size()) return nullptr;
    auto it = tab[layer].find(name);
    return it == tab[layer].end() ? nullptr : &it->second;
}

void free_table(std::vector<std::map<std::string, Tensor>> &tab) {
    for (auto &m : tab)
        for (auto &kv : m)
            if (kv.second.owned && kv.second.d) (void)hipFree(kv.second.d);
    tab.clear();
}

template <typename T>
int upload_array(dory_ctx *c, T **dst, const T *src, uint64_t n) {
    size_t b = n * sizeof(T);
    HIPCK(c, hipMalloc((void **)dst, b ? b : 256));
    if (b) HIPCK(c, hipMemcpy(*dst, src, b, hipMemcpyHostToDevice));
    return DORY_OK;
}

int ensure_scratch(dory_ctx *c, size_t bytes) {
    if (bytes <= c->scratch_bytes) return DORY_OK;
    if (c->capturing) return fail(c, DORY_ERR_ARG, "epoch graph: scratch would have to grow while recording (run one eager epoch first)");
    if (c->scratch) {
        HIPCK(c, hipStreamSynchronize(c->compute));
        (void)hipFree(c->scratch);
        c->scratch = nullptr;
        c->scratch_bytes = 0;
    }
    HIPCK(c, hipMalloc((void **)&c->scratch, bytes));
    c->scratch_bytes = bytes;
    return DORY_OK;
}

// longest-row-first schedule for skewed degree distributions
std::vector<uint32_t> degree_order(const uint64_t *ptr, uint32_t N) {
    std::vector<uint32_t> o(N);
    std::iota(o.begin(), o.end(), 0u);
    std::stable_sort(o.begin(), o.end(), [&](uint32_t a, uint32_t b) {
        return (ptr[a + 1] - ptr[a]) > (ptr[b + 1] - ptr[b]);
    });
    return o;
}

int gemm(dory_ctx *c, int ta, int tb, uint32_t M, uint32_t N, uint32_t K, const Tensor &A,
         const Tensor &B, Tensor &C, Tensor *C2 = nullptr) {
    GemmArgs g{};
    g.ta = ta; g.tb = tb; g.M = M; g.N = N; g.K = K;
    g.A = A.d; g.lda = A.ld; g.B = B.d; g.ldb = B.ld; g.C = C.d; g.ldc = C.ld;
    g.epilogue = C2 ? EPI_TANH : EPI_NONE;
    if (C2) { g.C2 = C2->d; g.ldc2 = C2->ld; }
    size_t need = gemm_scratch_bytes(M, N, K);
    if (need > ((size_t)256 << 20)) need = (size_t)256 << 20;
    int rc = ensure_scratch(c, need);
    if (rc) return rc;
    Timed t(c, "gemm", c->compute);
    HIPCK(c, launch_gemm(g, c->scratch, c->scratch_bytes, c->compute));
    return DORY_OK;
}
}  // namespace

static int ensure_blocked(dory_ctx *c, bool csc, int group);
static int blk_group_for(dory_ctx *c, uint32_t ld);

// Ghost rows of the last halo exchange land on the comm stream; with "halo_overlap" the
// compute stream is only made to wait for them (event ev_b) by the first consumer.
static int wait_halo(dory_ctx *c) {
    if (c->halo_pending) {
        HIPCK(c, hipStreamWaitEvent(c->compute, c->ev_b, 0));
        c->halo_pending = false;
    }
    return DORY_OK;
}

// Transform-first order for GCN layer 0 (opt-in, no reference counterpart): A(XW0) gathers d1-wide rows
// instead of the d0-wide rows of (AX)W0 -- 128 instead of 602 floats per edge on Reddit.  The weight gradient
// follows as X^T(A^T g0): one more d1-wide SpMM on the out-edges.  "ah"@0 is not produced in this mode.
static bool tf_active(dory_ctx *c) {
    return c->gnn == DORY_GCN && c->L >= 2 && c->dims[0] > c->dims[1] && c->opt["gcn_transform_first"] != 0 &&
           c->opt["adjacency_values_asymmetric"] == 0;
}

extern "C" {

// ---------------------------------------------------------------------------------------
int dory_create(int device, dory_ctx **out) {
    if (!out) return DORY_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return fail(nullptr, DORY_ERR_NODEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(nullptr, DORY_ERR_ARG, "device %d out of range (%d)", device, n);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess)
        return fail(nullptr, DORY_ERR_HIP, "hipGetDeviceProperties failed");
    if (!strstr(prop.gcnArchName, "gfx950"))
        return fail(nullptr, DORY_ERR_NODEVICE, "device %d is %s; kernels are built for gfx950 only", device,
                    prop.gcnArchName);
    dory_ctx *c = new dory_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->comm, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void **)&c->d_stat, 2 * sizeof(float)) != hipSuccess) {
        delete c;
        return fail(nullptr, DORY_ERR_HIP, "stream/event creation failed");
    }
    (void)hipMemset(c->d_stat, 0, 2 * sizeof(float));
    c->own_compute = c->own_comm = true;
    c->opt["spmm_variant"] = 1;      // 1: K1b source-blocked L2-resident gather where it applies, 0: K1 only
    c->opt["spmm_slab"] = 0;
    c->opt["spmm_order"] = 1;
    c->opt["spmm_blk_group"] = 32;   // K1b: lanes per row (slab = 4*group floats = 512 B)
    c->opt["spmm_blk_force_split"] = 0;   // testing: always launch local / ghost source blocks separately
    c->opt["halo_overlap"] = 1;      // let local-source blocks of the next SpMM run under the exchange
    c->opt["adjacency_values_asymmetric"] = 0;   // set by dory_partition_upload for undirected / unknown builds: csrVal != cscVal^T
    c->opt["gatmh_blocked"] = 1;         // multi-head GAT: source-blocked (L2-resident) gathers where the blocked adjacency applies
    c->opt["gcn_transform_first"] = 0;   // GCN layer 0 as A(XW) instead of (AX)W when the input is wider than the output (see tf_active)
    c->opt["epoch_graph"] = 0;       // engine: replay a recorded epoch (hipGraph) when the partition is alone
    c->opt["spmm_blk_nb"] = 0;       // K1b: number of source blocks (0 = auto, ~3.75 MB windows)
    *out = c;
    return DORY_OK;
}

static void free_graph(dory_ctx *c) {
    void *ps[] = {c->colPtr, c->rowPtr, c->rowIdx, c->colIdx, c->cscVal, c->csrVal, c->norm, c->orderIn, c->orderOut};
    for (void *p : ps)
        if (p) (void)hipFree(p);
    c->colPtr = c->rowPtr = nullptr;
    c->rowIdx = c->colIdx = nullptr;
    c->cscVal = c->csrVal = c->norm = nullptr;
    c->orderIn = c->orderOut = nullptr;
    free_blocked(&c->blkIn);
    free_blocked(&c->blkOut);
    c->blkIn_built = c->blkOut_built = false;
    c->blkIn_na = c->blkOut_na = false;
    c->has_graph = false;
}

int dory_destroy(dory_ctx *c) {
    if (!c) return DORY_ERR_ARG;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    drain_timing(c);
    for (auto &p : c->ev_pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    if (c->nccl) ncclCommDestroy((ncclComm_t)c->nccl);
    free_table(c->tensors);
    free_table(c->weights);
    free_table(c->wgrads);
    free_table(c->adam_m);
    free_table(c->adam_v);
    free_graph(c);
    for (int d = 0; d < 2; ++d) {
        if (c->plan[d].d_send_lvids) (void)hipFree(c->plan[d].d_send_lvids);
        if (c->plan[d].d_recv_slots) (void)hipFree(c->plan[d].d_recv_slots);
    }
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->partial) (void)hipFree(c->partial);
    if (c->epoch_exec) (void)hipGraphExecDestroy(c->epoch_exec);
    if (c->epoch_graph) (void)hipGraphDestroy(c->epoch_graph);
    if (c->d_lr_table) (void)hipFree(c->d_lr_table);
    if (c->d_replay_idx) (void)hipFree(c->d_replay_idx);
    if (c->send_buf) (void)hipFree(c->send_buf);
    if (c->recv_buf) (void)hipFree(c->recv_buf);
    if (c->d_stat) (void)hipFree(c->d_stat);
    if (c->ev_a) (void)hipEventDestroy(c->ev_a);
    if (c->ev_b) (void)hipEventDestroy(c->ev_b);
    if (c->own_compute && c->compute) (void)hipStreamDestroy(c->compute);
    if (c->own_comm && c->comm) (void)hipStreamDestroy(c->comm);
    delete c;
    return DORY_OK;
}

const char *dory_last_error(dory_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int dory_set_streams(dory_ctx *c, void *compute_stream, void *comm_stream) {
    CHECK_CTX(c);
    HIPCK(c, hipDeviceSynchronize());
    if (compute_stream) {
        if (c->own_compute) (void)hipStreamDestroy(c->compute);
        c->compute = (hipStream_t)compute_stream;
        c->own_compute = false;
    }
    if (comm_stream) {
        if (c->own_comm) (void)hipStreamDestroy(c->comm);
        c->comm = (hipStream_t)comm_stream;
        c->own_comm = false;
    }
    return DORY_OK;
}

int dory_sync(dory_ctx *c) {
    CHECK_CTX(c);
    HIPCK(c, hipStreamSynchronize(c->compute));
    HIPCK(c, hipStreamSynchronize(c->comm));
    c->halo_pending = false;   // everything has landed
    return DORY_OK;
}

int dory_configure(dory_ctx *c, int gnn_type, uint32_t num_layers, const uint32_t *dims,
                   uint32_t global_vtx_cnt, uint32_t node_id, uint32_t num_nodes) {
    CHECK_CTX(c);
    if (!dims || num_layers == 0 || (gnn_type != DORY_GCN && gnn_type != DORY_GAT && gnn_type != DORY_GATMH) || num_nodes == 0 ||
        node_id >= num_nodes)
        return fail(c, DORY_ERR_ARG, "dory_configure: bad arguments");
    for (uint32_t i = 0; i <= num_layers; ++i)
        if (dims[i] == 0) return fail(c, DORY_ERR_ARG, "dory_configure: zero layer width");
    c->gnn = gnn_type;
    c->L = num_layers;
    c->dims.assign(dims, dims + num_layers + 1);
    c->globalV = global_vtx_cnt;
    c->nodeId = node_id;
    c->numNodes = num_nodes;
    c->heads.assign(num_layers, 8);   // multi-head GAT extension defaults: 8 hidden heads, 1 output head
    c->heads[num_layers - 1] = 1;
    c->configured = true;
    return DORY_OK;
}

int dory_gatmh_heads(dory_ctx *c, const uint32_t *heads) {
    CHECK_CTX(c);
    if (!c->configured || c->gnn != DORY_GATMH || !heads) return fail(c, DORY_ERR_ARG, "gatmh_heads: configure with DORY_GATMH first");
    for (uint32_t l = 0; l < c->L; ++l)
        if (heads[l] == 0 || heads[l] > 64) return fail(c, DORY_ERR_ARG, "gatmh_heads: bad head count");
    c->heads.assign(heads, heads + c->L);
    return DORY_OK;
}

int dory_graph_upload(dory_ctx *c, uint32_t N, uint32_t Gsrc, uint32_t Gdst, uint64_t nnz_in,
                      const uint64_t *column_ptrs, const uint32_t *row_idxs, const float *csc_values,
                      uint64_t nnz_out, const uint64_t *row_ptrs, const uint32_t *column_idxs,
                      const float *csr_values, const float *vtx_norms) {
    CHECK_CTX(c);
    if (!column_ptrs || !row_ptrs || (N && !vtx_norms) || (nnz_in && (!row_idxs || !csc_values)) ||
        (nnz_out && (!column_idxs || !csr_values)))
        return fail(c, DORY_ERR_ARG, "dory_graph_upload: null array");
    if (column_ptrs[0] != 0 || column_ptrs[N] != nnz_in || row_ptrs[0] != 0 || row_ptrs[N] != nnz_out)
        return fail(c, DORY_ERR_ARG, "dory_graph_upload: pointer arrays do not match nnz");
    for (uint32_t v = 0; v < N; ++v)
        if (column_ptrs[v] > column_ptrs[v + 1] || row_ptrs[v] > row_ptrs[v + 1])
            return fail(c, DORY_ERR_ARG, "dory_graph_upload: pointer array not monotone at %u", v);
    for (uint64_t e = 0; e < nnz_in; ++e)
        if (row_idxs[e] >= (uint64_t)N + Gsrc) return fail(c, DORY_ERR_ARG, "row index %u out of range at %llu", row_idxs[e], (unsigned long long)e);
    for (uint64_t e = 0; e < nnz_out; ++e)
        if (column_idxs[e] >= (uint64_t)N + Gdst) return fail(c, DORY_ERR_ARG, "column index %u out of range at %llu", column_idxs[e], (unsigned long long)e);
    HIPCK(c, hipDeviceSynchronize());
    free_graph(c);
    c->N = N; c->Gsrc = Gsrc; c->Gdst = Gdst; c->nnz_in = nnz_in; c->nnz_out = nnz_out;
    int rc;
    if ((rc = upload_array(c, &c->colPtr, column_ptrs, (uint64_t)N + 1))) return rc;
    if ((rc = upload_array(c, &c->rowIdx, row_idxs, nnz_in))) return rc;
    if ((rc = upload_array(c, &c->cscVal, csc_values, nnz_in))) return rc;
    if ((rc = upload_array(c, &c->rowPtr, row_ptrs, (uint64_t)N + 1))) return rc;
    if ((rc = upload_array(c, &c->colIdx, column_idxs, nnz_out))) return rc;
    if ((rc = upload_array(c, &c->csrVal, csr_values, nnz_out))) return rc;
    if ((rc = upload_array(c, &c->norm, vtx_norms, (uint64_t)N))) return rc;
    auto oi = degree_order(column_ptrs, N), oo = degree_order(row_ptrs, N);
    if ((rc = upload_array(c, &c->orderIn, oi.data(), (uint64_t)N))) return rc;
    if ((rc = upload_array(c, &c->orderOut, oo.data(), (uint64_t)N))) return rc;
    c->has_graph = true;
    return DORY_OK;
}

int dory_preallocate(dory_ctx *c) {
    CHECK_CTX(c);
    if (!c->configured || !c->has_graph) return fail(c, DORY_ERR_ARG, "dory_preallocate: configure and graph_upload first");
    HIPCK(c, hipDeviceSynchronize());
    free_table(c->tensors); free_table(c->weights); free_table(c->wgrads); free_table(c->adam_m); free_table(c->adam_v);
    const uint32_t L = c->L, N = c->N;
    auto &d = c->dims;
    c->tensors.assign(L + 1, {});
    c->weights.assign(L, {}); c->wgrads.assign(L, {}); c->adam_m.assign(L, {}); c->adam_v.assign(L, {});
    int rc = 0;
    auto mk = [&](uint32_t layer, const char *name, uint64_t rows, uint32_t cols) {
        if (rc) return;
        rc = alloc_tensor(c, c->tensors[layer][name], rows, cols);
    };
    if (c->gnn == DORY_GCN) {  // Engine::preallocateGCN (engine/ops/gcn_ops.cpp:27-93)
        mk(0, "x", N, d[0]);
        mk(0, "fg", c->Gsrc, d[0]);
        mk(L - 1, "lab", N, d[L]);
        for (uint32_t l = 0; l < L; ++l) {
            mk(l, "ah", N, d[l]);
            mk(l, "z", N, d[l + 1]);            // reference keeps z only for l < L-1; last-layer logits are a temporary there
            if (l < L - 1) {
                mk(l, "h", N, d[l + 1]);
                mk(l + 1, "fg", c->Gsrc, d[l + 1]);
            }
            mk(l, "g", N, d[l + 1]);            // interGrad / d_output temporaries of CPU_comm.cpp:121,143
        }
        for (uint32_t l = L - 1; l > 0; --l) {
            mk(l, "grad", N, d[l]);
            mk(l - 1, "bg", c->Gdst, d[l]);
            mk(l - 1, "aTg", N, d[l]);
        }
        if (L >= 2 && d[0] > d[1]) {   // transform-first order of layer 0 (option gcn_transform_first)
            mk(0, "xw", N, d[1]);          // X W0
            mk(0, "fgxw", c->Gsrc, d[1]);  // the same for the layer-0 ghost rows
            mk(0, "u", N, d[1]);           // A^T g0
            mk(0, "bgg", c->Gdst, d[1]);   // ghost rows of g0 (backward exchange at layer 0)
        }
    } else if (c->gnn == DORY_GATMH) {  // extension (no reference counterpart): see dory_gatmh_heads
        if (c->numNodes > 1) return fail(c, DORY_ERR_ARG, "multi-head GAT extension: single partition only in this version");
        mk(0, "h", N, d[0]);
        mk(L - 1, "lab", N, d[L]);
        mk(L - 1, "logits", N, d[L]);
        mk(L - 1, "grad", N, d[L]);
        for (uint32_t l = 0; l < L; ++l) {
            const uint32_t K = c->heads[l];
            const bool last = l == L - 1;
            const uint32_t zw = last ? d[l + 1] * K : d[l + 1];
            const uint32_t D = zw / K;
            if (zw % K || zw > 256 || (K > 1 && ((D & (D - 1)) || D > 64)))
                return fail(c, DORY_ERR_ARG, "multi-head GAT: layer %u width %u does not split into %u heads (D power of two <= 64, K*D <= 256)", l, zw, K);
            mk(l, "z", N, zw);
            mk(l, "o", N, zw);
            mk(l, "do", N, zw);
            mk(l, "dz", N, zw);
            for (const char *nm : {"el", "er", "m", "den", "t", "del", "der"}) mk(l, nm, N, K);
            if (!last) mk(l + 1, "h", N, d[l + 1]);
            if (l > 0) mk(l, "dh", N, d[l]);
        }
    } else {  // Engine::preallocateGAT (engine/ops/gat_ops.cpp:27-115)
        mk(0, "h", N, d[0]);
        mk(L - 1, "lab", N, d[L]);
        for (uint32_t l = 0; l < L; ++l) {
            mk(l, "z", N, d[l + 1]);
            mk(l, "az", c->nnz_in, 1);
            mk(l, "fg_z", c->Gsrc, d[l + 1]);
            Tensor A;  // "A" aliases forwardAdj.values (gat_ops.cpp:61-64)
            A.rows = c->nnz_in; A.cols = 1; A.ld = 1; A.d = c->cscVal; A.owned = false;
            c->tensors[l]["A"] = A;
            mk(l, "ah", N, d[l + 1]);
            if (l < L - 1) mk(l + 1, "h", N, d[l + 1]);
            mk(l, "grad", N, d[l + 1]);
            mk(l, "dA", c->nnz_in, 1);
            mk(l, "aTg", N, d[l + 1]);
            mk(l, "bg_d", c->Gdst, d[l + 1]);
        }
        mk(0, "cw", N, 1);  // column weights for the a_i gradient (K5)
        for (uint32_t l = 0; l < L; ++l) {
            mk(l, "arow", N, 1);   // per-destination value of "A"  (all edges of a column are equal)
            mk(l, "drow", N, 1);   // per-destination value of "dA"
        }
        c->gat_arow_valid.assign(L, 0);
        c->gat_drow_valid.assign(L, 0);
    }
    if (rc) return rc;
    for (uint32_t l = 0; l < L; ++l) {
        if (c->gnn == DORY_GATMH) {
            const uint32_t zw = l == L - 1 ? d[l + 1] * c->heads[l] : d[l + 1];
            for (auto *tab : {&c->weights, &c->wgrads, &c->adam_m, &c->adam_v}) {
                if ((rc = alloc_tensor(c, (*tab)[l]["w"], d[l], zw))) return rc;
                if ((rc = alloc_tensor(c, (*tab)[l]["a_l"], zw, 1))) return rc;
                if ((rc = alloc_tensor(c, (*tab)[l]["a_r"], zw, 1))) return rc;
            }
            continue;
        }
        if ((rc = alloc_tensor(c, c->weights[l]["w"], d[l], d[l + 1]))) return rc;
        if ((rc = alloc_tensor(c, c->wgrads[l]["w"], d[l], d[l + 1]))) return rc;
        if ((rc = alloc_tensor(c, c->adam_m[l]["w"], d[l], d[l + 1]))) return rc;
        if ((rc = alloc_tensor(c, c->adam_v[l]["w"], d[l], d[l + 1]))) return rc;
        if (c->gnn == DORY_GAT) {
            if ((rc = alloc_tensor(c, c->weights[l]["a_i"], d[l + 1], 1))) return rc;
            if ((rc = alloc_tensor(c, c->wgrads[l]["a_i"], d[l + 1], 1))) return rc;
            if ((rc = alloc_tensor(c, c->adam_m[l]["a_i"], d[l + 1], 1))) return rc;
            if ((rc = alloc_tensor(c, c->adam_v[l]["a_i"], d[l + 1], 1))) return rc;
        }
    }
    c->adam.epochs = 1;
    HIPCK(c, hipStreamSynchronize(c->compute));
    if (c->opt["spmm_variant"] == 1 && N > 0 && c->gnn == DORY_GATMH && c->opt["gatmh_blocked"]) {
        // the extension's forward sum gathers through the same source-blocked copy of the in-edges
        uint32_t maxld = 0;
        for (uint32_t l = 0; l < L; ++l) maxld = std::max(maxld, pad_ld(l == L - 1 ? d[l + 1] * c->heads[l] : d[l + 1]));
        if ((rc = ensure_blocked(c, true, blk_group_for(c, maxld)))) return rc;
        if ((rc = ensure_blocked(c, false, blk_group_for(c, maxld)))) return rc;   // backward, source side
        const uint32_t nbmax = std::max(c->blkIn.nb, c->blkOut.nb);
        const size_t need = (size_t)nbmax * N * (maxld + 64) * sizeof(float);      // + per-(block,row,head) partials
        if (nbmax && need <= ((size_t)48 << 30) && need > c->partial_bytes) {
            if (c->partial) (void)hipFree(c->partial);
            c->partial = nullptr;
            c->partial_bytes = 0;
            HIPCK(c, hipMalloc((void **)&c->partial, need));
            c->partial_bytes = need;
        }
    }
    if (c->opt["spmm_variant"] == 1 && N > 0 && c->gnn != DORY_GATMH) {   // K1b: regroup the edges now, not inside the first epoch
        uint32_t minld = 0xFFFFFFFFu;
        for (uint32_t l = 0; l < L; ++l) {
            const uint32_t w = c->gnn == DORY_GCN ? (l == 0 ? d[0] : d[l]) : d[l + 1];
            minld = std::min(minld, pad_ld(w));
        }
        uint32_t maxld = 0;
        for (uint32_t l = 0; l <= L; ++l) maxld = std::max(maxld, pad_ld(d[l]));
        const int group = blk_group_for(c, maxld);   // block size for the widest rows (most of the traffic)
        if (minld >= 32) {
            if ((rc = ensure_blocked(c, true, group))) return rc;
            if ((rc = ensure_blocked(c, false, group))) return rc;
            // the partial-sum buffer too, so that no allocation happens inside an epoch
            const uint32_t nbmax = std::max(c->blkIn.nb, c->blkOut.nb);
            const size_t need = (size_t)nbmax * N * maxld * sizeof(float);
            if (nbmax && need <= ((size_t)48 << 30) && need > c->partial_bytes) {
                if (c->partial) (void)hipFree(c->partial);
                c->partial = nullptr;
                c->partial_bytes = 0;
                HIPCK(c, hipMalloc((void **)&c->partial, need));
                c->partial_bytes = need;
            }
        }
    }
    c->prealloc = true;
    return DORY_OK;
}

// ---------------------------------------------------------------------------------------
int dory_tensor_info(dory_ctx *c, uint32_t layer, const char *name, uint64_t *rows, uint32_t *cols,
                     uint32_t *ld, void **device_ptr) {
    CHECK_CTX(c);
    Tensor *t = name ? find(c, layer, name) : nullptr;
    if (!t) return fail(c, DORY_ERR_ARG, "no tensor '%s' at layer %u", name ? name : "(null)", layer);
    if (rows) *rows = t->rows;
    if (cols) *cols = t->cols;
    if (ld) *ld = t->ld;
    if (device_ptr) *device_ptr = t->d;
    return DORY_OK;
}

static int upload_dense(dory_ctx *c, Tensor &t, const float *host) {
    if (t.rows == 0 || t.cols == 0) return DORY_OK;
    if (t.ld == t.cols) {
        HIPCK(c, hipMemcpyAsync(t.d, host, t.bytes(), hipMemcpyHostToDevice, c->compute));
    } else {
        float *stage = nullptr;
        const size_t b = (size_t)t.rows * t.cols * sizeof(float);
        HIPCK(c, hipMalloc((void **)&stage, b));
        HIPCK(c, hipMemcpyAsync(stage, host, b, hipMemcpyHostToDevice, c->compute));
        HIPCK(c, launch_pad_copy(t.d, t.ld, stage, t.cols, t.rows, t.cols, c->compute));
        HIPCK(c, hipStreamSynchronize(c->compute));
        (void)hipFree(stage);
    }
    HIPCK(c, hipStreamSynchronize(c->compute));
    return DORY_OK;
}

static int download_dense(dory_ctx *c, const Tensor &t, float *host) {
    if (t.rows == 0 || t.cols == 0) return DORY_OK;
    HIPCK(c, hipStreamSynchronize(c->comm));
    if (t.ld == t.cols) {
        HIPCK(c, hipMemcpyAsync(host, t.d, t.bytes(), hipMemcpyDeviceToHost, c->compute));
    } else {
        HIPCK(c, hipMemcpy2DAsync(host, (size_t)t.cols * sizeof(float), t.d, (size_t)t.ld * sizeof(float),
                                  (size_t)t.cols * sizeof(float), t.rows, hipMemcpyDeviceToHost, c->compute));
    }
    HIPCK(c, hipStreamSynchronize(c->compute));
    return DORY_OK;
}

int dory_tensor_upload(dory_ctx *c, uint32_t layer, const char *name, const float *host) {
    CHECK_CTX(c);
    { int wrc = wait_halo(c); if (wrc) return wrc; }
    Tensor *t = name ? find(c, layer, name) : nullptr;
    if (!t || !host) return fail(c, DORY_ERR_ARG, "tensor_upload: no tensor '%s' at layer %u", name ? name : "(null)", layer);
    if (!strcmp(name, "A")) for (auto &f : c->gat_arow_valid) f = 0;          // caller-supplied edge weights: general path
    if (!strcmp(name, "dA") && layer < c->gat_drow_valid.size()) c->gat_drow_valid[layer] = 0;
    return upload_dense(c, *t, host);
}

int dory_tensor_download(dory_ctx *c, uint32_t layer, const char *name, float *host) {
    CHECK_CTX(c);
    { int wrc = wait_halo(c); if (wrc) return wrc; }
    Tensor *t = name ? find(c, layer, name) : nullptr;
    if (!t || !host) return fail(c, DORY_ERR_ARG, "tensor_download: no tensor '%s' at layer %u", name ? name : "(null)", layer);
    return download_dense(c, *t, host);
}

int dory_tensor_fill_uniform(dory_ctx *c, uint32_t layer, const char *name, uint64_t seed, float lo,
                             float hi, const uint32_t *global_row_ids) {
    CHECK_CTX(c);
    { int wrc = wait_halo(c); if (wrc) return wrc; }
    Tensor *t = name ? find(c, layer, name) : nullptr;
    if (!t) return fail(c, DORY_ERR_ARG, "tensor_fill: no tensor '%s' at layer %u", name ? name : "(null)", layer);
    uint32_t *ids = nullptr;
    if (global_row_ids && t->rows) {
        int rc = upload_array(c, &ids, global_row_ids, t->rows);
        if (rc) return rc;
    }
    HIPCK(c, launch_fill_uniform_ids(t->d, t->rows, t->cols, t->ld, ids, seed, lo, hi, c->compute));
    HIPCK(c, hipStreamSynchronize(c->compute));
    if (ids) (void)hipFree(ids);
    return DORY_OK;
}

int dory_labels_upload(dory_ctx *c, const uint32_t *labels) {
    CHECK_CTX(c);
    if (!c->prealloc) return fail(c, DORY_ERR_ARG, "labels_upload: preallocate first");
    Tensor *t = find(c, c->L - 1, "lab");
    if (!t || (!labels && t->rows)) return fail(c, DORY_ERR_ARG, "labels_upload: bad arguments");
    for (uint64_t i = 0; i < t->rows; ++i)
        if (labels[i] >= t->cols) return fail(c, DORY_ERR_ARG, "label %u at row %llu exceeds %u classes", labels[i], (unsigned long long)i, t->cols);
    uint32_t *dl = nullptr;
    int rc = upload_array(c, &dl, labels, t->rows);
    if (rc) return rc;
    HIPCK(c, launch_onehot(t->d, t->rows, t->cols, t->ld, dl, c->compute));
    HIPCK(c, hipStreamSynchronize(c->compute));
    (void)hipFree(dl);
    return DORY_OK;
}

// ---------------------------------------------------------------------------------------
int dory_weight_set(dory_ctx *c, uint32_t layer, const char *name, const float *host) {
    CHECK_CTX(c);
    Tensor *t = name ? findw(c->weights, layer, name) : nullptr;
    if (!t || !host) return fail(c, DORY_ERR_ARG, "weight_set: no weight '%s' at layer %u", name ? name : "(null)", layer);
    return upload_dense(c, *t, host);
}
int dory_weight_get(dory_ctx *c, uint32_t layer, const char *name, float *host) {
    CHECK_CTX(c);
    Tensor *t = name ? findw(c->weights, layer, name) : nullptr;
    if (!t || !host) return fail(c, DORY_ERR_ARG, "weight_get: no weight '%s' at layer %u", name ? name : "(null)", layer);
    return download_dense(c, *t, host);
}
int dory_weight_grad_get(dory_ctx *c, uint32_t layer, const char *name, float *host) {
    CHECK_CTX(c);
    Tensor *t = name ? findw(c->wgrads, layer, name) : nullptr;
    if (!t || !host) return fail(c, DORY_ERR_ARG, "weight_grad_get: no gradient '%s' at layer %u", name ? name : "(null)", layer);
    return download_dense(c, *t, host);
}

int dory_weights_init_xavier(dory_ctx *c) {
    CHECK_CTX(c);
    if (!c->prealloc) return fail(c, DORY_ERR_ARG, "weights_init: preallocate first");
    for (uint32_t l = 0; l < c->L; ++l) {
        // WeightServer::xavierInitializer (src/weight-server/weightserver.cpp:567-585):
        // every tensor restarts the engine at seed 8888.
        for (auto &kv : c->weights[l]) {
            Tensor &t = kv.second;
            const uint32_t d1 = (uint32_t)t.rows, d2 = t.cols;
            std::vector<float> w((size_t)d1 * d2);
            std::default_random_engine dre(8888);
            std::uniform_real_distribution<float> dist(-1, 1);
            for (auto &x : w) x = dist(dre);
            const float nf = std::sqrt(6.0 / (float(d1 + d2)));
            for (auto &x : w) x *= nf;
            int rc = upload_dense(c, t, w.data());
            if (rc) return rc;
        }
    }
    return DORY_OK;
}

// ---------------------------------------------------------------------------------------
// K1b bookkeeping: (re)build the source-blocked copy of one adjacency for `group` lanes/row
static int ensure_blocked(dory_ctx *c, bool csc, int group) {
    BlockedAdj &B = csc ? c->blkIn : c->blkOut;
    bool &built = csc ? c->blkIn_built : c->blkOut_built;
    const uint32_t want_nb = (uint32_t)c->opt["spmm_blk_nb"];
    // the block structure serves every slab width; only an explicit block count forces a rebuild
    if (c->capturing && (!built || (want_nb && B.nb != (want_nb + 7) / 8 * 8)) && !(csc ? c->blkIn_na : c->blkOut_na))
        return fail(c, DORY_ERR_ARG, "epoch graph: blocked adjacency would have to be (re)built while recording");
    if (built && want_nb && B.nb != (want_nb + 7) / 8 * 8) {
        HIPCK(c, hipStreamSynchronize(c->compute));
        free_blocked(&B);
        built = false;
    }
    if (!built) {
        const uint32_t NG = c->N + (csc ? c->Gsrc : c->Gdst);
        // K1b pays nb partial rows per output row: only worth it (and only affordable: the
        // per-(block,row) offset table is nb*(N+1) words) while the source space is a few
        // hundred L2 windows at most.  Larger partitions keep K1.
        const uint32_t nb = plan_blocks(NG, want_nb, (uint32_t)group * 16u);
        // ... and pointless when the whole source slab fits one XCD's L2 anyway (Cora-sized graphs):
        // K1 then gathers from L2 without partial sums or a second kernel
        const bool tiny = !want_nb && (uint64_t)NG * group * 16u <= ((uint64_t)4 << 20);
        if (tiny || nb > 256 || (uint64_t)nb * (c->N + 1) * 8ull > ((uint64_t)8 << 30)) {
            (csc ? c->blkIn_na : c->blkOut_na) = true;
            return DORY_OK;
        }
        HIPCK(c, build_blocked(csc ? c->colPtr : c->rowPtr, csc ? c->rowIdx : c->colIdx, csc ? c->cscVal : c->csrVal,
                               c->N, NG, csc ? c->nnz_in : c->nnz_out, want_nb, (uint32_t)group * 16u, &B, c->compute));
        B.row_bytes = (uint32_t)group * 16u;
        built = true;
    }
    return DORY_OK;
}

static int blk_group_for(dory_ctx *c, uint32_t ld) {
    int group = (int)c->opt["spmm_blk_group"];
    if (group != 8 && group != 16 && group != 32) group = 32;
    if (ld < 128 && group == 32) group = 16;   // narrow tensors: one 256-B slab
    return group;
}

// One aggregation.  Edge weights come from `val` (any per-edge array, K1), or -- when
// `val` is the adjacency's own static array -- from the source-blocked copy (K1b), or are
// 1 with a per-destination factor `row_scale` (K1b, unit mode; the reference GAT's edge
// scores depend on the destination only, CPU_comm.cpp:299-319).
static int spmm(dory_ctx *c, bool csc, const float *val, int self_mode, Tensor &xl, Tensor *xg, Tensor &out,
                uint32_t F, int accumulate, const float *row_scale = nullptr) {
    if (xl.ld != out.ld || (xg && xg->rows && xg->ld != xl.ld) || xl.cols != F)
        return fail(c, DORY_ERR_ARG, "spmm: tensor shapes disagree (F=%u ld %u/%u)", F, xl.ld, out.ld);
    SpmmArgs a{};
    a.N = c->N; a.F = F; a.ld = xl.ld;
    a.ptr = csc ? c->colPtr : c->rowPtr;
    a.idx = csc ? c->rowIdx : c->colIdx;
    a.val = val;
    a.self_scale = c->norm;
    a.self_mode = self_mode;
    a.xl = xl.d; a.xg = xg ? xg->d : nullptr; a.out = out.d;
    a.accumulate = accumulate;
    a.order = c->opt["spmm_order"] ? (csc ? c->orderIn : c->orderOut) : nullptr;
    const bool static_vals = val == (csc ? c->cscVal : c->csrVal) && !(c->gnn == DORY_GAT && csc);  // GAT rewrites cscVal
    if (c->opt["spmm_variant"] == 1 && (static_vals || row_scale) && c->N > 0 && a.ld >= 32) {
        const int group = blk_group_for(c, a.ld);
        int rc = ensure_blocked(c, csc, group);
        if (rc) return rc;
        BlockedAdj &B = csc ? c->blkIn : c->blkOut;
        const size_t need = blocked_partial_bytes(a, B);
        if (!(csc ? c->blkIn_na : c->blkOut_na) && B.nb > 0 && need <= ((size_t)48 << 30)) {
            if (need > c->partial_bytes) {
                if (c->capturing) return fail(c, DORY_ERR_ARG, "epoch graph: partial buffer would have to grow while recording");
                HIPCK(c, hipStreamSynchronize(c->compute));
                if (c->partial) (void)hipFree(c->partial);
                c->partial = nullptr;
                c->partial_bytes = 0;
                HIPCK(c, hipMalloc((void **)&c->partial, need));
                c->partial_bytes = need;
            }
            Timed t(c, "spmm", c->compute);
            // source blocks that contain local rows only do not depend on the exchange in
            // flight: they run first, the ghost blocks after the comm stream's event
            const uint32_t nb_local = std::min(B.nb, c->N / B.SB);
            const bool split = (c->halo_pending || c->opt["spmm_blk_force_split"]) && nb_local > 0 && nb_local < B.nb;
            if (split) {
                HIPCK(c, launch_spmm_blocked_part(a, B, c->partial, group, row_scale != nullptr, 0, nb_local, c->compute));
                if ((rc = wait_halo(c))) return rc;
                HIPCK(c, launch_spmm_blocked_part(a, B, c->partial, group, row_scale != nullptr, nb_local, B.nb, c->compute));
            } else {
                if ((rc = wait_halo(c))) return rc;
                HIPCK(c, launch_spmm_blocked_part(a, B, c->partial, group, row_scale != nullptr, 0, B.nb, c->compute));
            }
            HIPCK(c, launch_spmm_blocked_reduce(a, B, c->partial, row_scale, c->compute));
            return DORY_OK;
        }
    }
    if (!val) return fail(c, DORY_ERR_ARG, "spmm: no edge values");
    {
        int rc = wait_halo(c);
        if (rc) return rc;
    }
    Timed t(c, "spmm", c->compute);
    HIPCK(c, launch_spmm(a, (int)c->opt["spmm_variant"], (int)c->opt["spmm_slab"], c->compute));
    return DORY_OK;
}

#define NEED(ptr, l, nm)                                                                  \
    Tensor *ptr = find(c, (l), nm);                                                       \
    if (!ptr) return fail(c, DORY_ERR_ARG, "%s: tensor '%s'@%u missing", __func__, nm, (unsigned)(l))

int dory_aggregate(dory_ctx *c, uint32_t layer, int dir) {
    CHECK_CTX(c);
    if (!c->prealloc) return fail(c, DORY_ERR_ARG, "aggregate: preallocate first");
    if (c->gnn == DORY_GCN) {  // Engine::aggregateGCN (gcn_ops.cpp:130-191)
        if (dir == DORY_FORWARD) {
            if (layer >= c->L) return fail(c, DORY_ERR_ARG, "aggregate: layer %u out of range", layer);
            Tensor *in = layer == 0 ? find(c, 0, "x") : find(c, layer - 1, "h");
            NEED(fg, layer, "fg");
            NEED(ah, layer, "ah");
            if (!in) return fail(c, DORY_ERR_ARG, "aggregate: input tensor missing");
            if (layer == 0 && tf_active(c)) {   // z0 = A (X W0): transform the local and the ghost rows, then gather d1-wide
                NEED(xw, 0, "xw"); NEED(fgxw, 0, "fgxw"); NEED(z, 0, "z");
                Tensor &W = c->weights[0]["w"];
                int rc = gemm(c, 0, 0, c->N, c->dims[1], c->dims[0], *in, W, *xw);
                if (!rc && c->Gsrc) rc = gemm(c, 0, 0, c->Gsrc, c->dims[1], c->dims[0], *fg, W, *fgxw);
                if (rc) return rc;
                return spmm(c, true, c->cscVal, 1, *xw, fgxw, *z, c->dims[1], 0);
            }
            return spmm(c, true, c->cscVal, 1, *in, fg, *ah, c->dims[layer], 0);
        }
        if (layer == 0 && tf_active(c)) {   // dW0 = X^T (A^T g0)   (ghost rows of g0: halo exchange (0, backward))
            NEED(g, 0, "g"); NEED(bgg, 0, "bgg"); NEED(u, 0, "u"); NEED(x, 0, "x");
            int rc = spmm(c, false, c->csrVal, 1, *g, bgg, *u, c->dims[1], 0);
            if (rc) return rc;
            return gemm(c, 1, 0, c->dims[0], c->dims[1], c->N, *x, *u, c->wgrads[0]["w"]);
        }
        if (layer == 0 || layer >= c->L) return fail(c, DORY_ERR_ARG, "aggregate backward: layer %u out of range", layer);
        NEED(grad, layer, "grad");
        NEED(bg, layer - 1, "bg");
        NEED(aTg, layer - 1, "aTg");
        return spmm(c, false, c->csrVal, 1, *grad, bg, *aTg, c->dims[layer], 0);
    }
    // Engine::aggregateGAT (gat_ops.cpp:173-243): tensors live at layer-1
    if (layer == 0 || layer > c->L) return fail(c, DORY_ERR_ARG, "aggregate GAT: layer %u out of range", layer);
    const uint32_t fl = layer - 1;
    if (c->gnn == DORY_GATMH) {  // extension: edge softmax + weighted sum, and its backward
        const uint32_t K = c->heads[fl];
        const bool last = fl == c->L - 1;
        NEED(z, fl, "z"); NEED(el, fl, "el"); NEED(er, fl, "er"); NEED(m, fl, "m"); NEED(den, fl, "den"); NEED(o, fl, "o");
        const uint32_t D = z->cols / K;
        if (dir == DORY_FORWARD) {
            {
                Timed t(c, "spmm", c->compute);
                const bool blocked = c->opt["gatmh_blocked"] && c->blkIn_built && !c->blkIn_na && c->blkIn.nb > 0 &&
                                     (D % 4 == 0 || K == 1) &&
                                     (size_t)c->blkIn.nb * c->N * z->ld * sizeof(float) <= c->partial_bytes;
                if (blocked)
                    HIPCK(c, launch_gatmh_forward_blocked(c->N, K, D, z->ld, el->ld, c->colPtr, c->rowIdx, c->blkIn, z->d,
                                                          el->d, er->d, o->d, m->d, den->d, c->partial, c->compute));
                else
                    HIPCK(c, launch_gatmh_forward(c->N, K, D, z->ld, el->ld, c->colPtr, c->rowIdx, z->d, el->d, er->d,
                                                  o->d, m->d, den->d, c->compute));
            }
            Timed t(c, "loss", c->compute);
            if (!last) {
                NEED(hn, fl + 1, "h");
                HIPCK(c, launch_gatmh_elu(c->N, o->cols, o->d, o->ld, hn->d, hn->ld, c->compute));
            } else {
                NEED(lg, fl, "logits");
                HIPCK(c, launch_gatmh_head_mean(c->N, K, lg->cols, o->d, o->ld, lg->d, lg->ld, c->compute));
            }
            return DORY_OK;
        }
        NEED(dO, fl, "do"); NEED(dz, fl, "dz"); NEED(tt, fl, "t"); NEED(del, fl, "del"); NEED(der, fl, "der");
        {
            Timed t(c, "loss", c->compute);
            if (last) {
                NEED(gr, fl, "grad");
                HIPCK(c, launch_gatmh_head_expand(c->N, K, gr->cols, gr->d, gr->ld, dO->d, dO->ld, c->compute));
            } else {
                NEED(dh, fl + 1, "dh");
                HIPCK(c, launch_gatmh_elu_bwd(c->N, o->cols, dh->d, dh->ld, o->d, o->ld, dO->d, dO->ld, c->compute));
            }
        }
        int rc = ensure_scratch(c, (size_t)1024 * z->cols * sizeof(float) + (size_t)c->N * K * 16 + 256);
        if (rc) return rc;
        Timed t(c, "spmm", c->compute);
        const uint32_t nbmax = std::max(c->blkIn.nb, c->blkOut.nb);
        if (c->opt["gatmh_blocked"] && c->blkIn_built && c->blkOut_built && !c->blkIn_na && !c->blkOut_na && nbmax > 0 &&
            gatmh_backward_blocked_ok(K, D, z->ld) &&
            (size_t)nbmax * c->N * (z->ld + K) * sizeof(float) <= c->partial_bytes &&
            c->scratch_bytes >= (size_t)c->N * K * 16 + 256 + (size_t)z->cols * sizeof(float)) {
            float4 *st4 = reinterpret_cast<float4 *>(c->scratch);
            const size_t st4_bytes = ((size_t)c->N * K * 16 + 255) & ~(size_t)255;
            HIPCK(c, launch_gatmh_backward_blocked(c->N, K, D, z->ld, el->ld, c->blkIn, c->blkOut, z->d, el->d, er->d, m->d,
                                                   den->d, dO->d, c->weights[fl]["a_l"].d, c->weights[fl]["a_r"].d, tt->d,
                                                   del->d, der->d, dz->d, c->partial, st4, c->compute));
            HIPCK(c, launch_gatmh_dattn(c->N, K, D, z->ld, el->ld, z->d, del->d, der->d, c->wgrads[fl]["a_l"].d,
                                        c->wgrads[fl]["a_r"].d, c->scratch + st4_bytes / sizeof(float),
                                        c->scratch_bytes - st4_bytes, c->compute));
            return DORY_OK;
        }
        HIPCK(c, launch_gatmh_backward(c->N, K, D, z->ld, el->ld, c->colPtr, c->rowIdx, c->rowPtr, c->colIdx, z->d, el->d,
                                       er->d, m->d, den->d, dO->d, c->weights[fl]["a_l"].d, c->weights[fl]["a_r"].d,
                                       tt->d, del->d, der->d, dz->d, c->wgrads[fl]["a_l"].d, c->wgrads[fl]["a_r"].d,
                                       c->scratch, c->scratch_bytes, c->compute));
        return DORY_OK;
    }
    NEED(z, fl, "z");
    NEED(fgz, fl, "fg_z");
    // dory_apply_edge leaves, next to the per-edge tensors "A" / "dA", the one value all
    // edges of a destination share; while that is current the SpMM gathers unweighted
    // (K1b) and scales per row.  A caller that overwrote "A"/"dA" gets the general K1 path.
    if (dir == DORY_FORWARD) {
        NEED(ah, fl, "ah");
        Tensor *arow = find(c, fl, "arow");
        const bool fast = arow && fl < c->gat_arow_valid.size() && c->gat_arow_valid[fl];
        return spmm(c, true, c->cscVal, 2, *z, fgz, *ah, c->dims[layer], 0, fast ? arow->d : nullptr);
    }
    NEED(grad, fl, "grad");
    NEED(bgd, fl, "bg_d");
    NEED(dA, fl, "dA");
    NEED(aTg, fl, "aTg");
    // fresh two-term sum (the CUDA path's semantics, gat_ops.cpp:155-163): A^T.dP then += dA.Z
    int rc = spmm(c, false, c->csrVal, 0, *grad, bgd, *aTg, c->dims[layer], 0);
    if (rc) return rc;
    Tensor *drow = find(c, fl, "drow");
    const bool fast = drow && fl < c->gat_drow_valid.size() && c->gat_drow_valid[fl];
    return spmm(c, true, dA->d, 0, *z, fgz, *aTg, c->dims[layer], 1, fast ? drow->d : nullptr);
}

int dory_apply_vertex(dory_ctx *c, uint32_t layer, int dir) {
    CHECK_CTX(c);
    { int wrc = wait_halo(c); if (wrc) return wrc; }
    if (!c->prealloc) return fail(c, DORY_ERR_ARG, "apply_vertex: preallocate first");
    if (layer >= c->L) return fail(c, DORY_ERR_ARG, "apply_vertex: layer %u out of range", layer);
    const uint32_t N = c->N, Fin = c->dims[layer], Fout = c->dims[layer + 1];
    Tensor &W = c->weights[layer]["w"];
    Tensor &dW = c->wgrads[layer]["w"];
    int rc;
    if (c->gnn == DORY_GCN) {
        NEED(ah, layer, "ah");
        NEED(z, layer, "z");
        NEED(g, layer, "g");
        if (dir == DORY_FORWARD) {
            if (layer != c->L - 1) {  // vtxNNForwardGCN hidden (CPU_comm.cpp:98-107)
                NEED(h, layer, "h");
                if (layer == 0 && tf_active(c)) {   // z0 came out of dory_aggregate already
                    Timed t(c, "loss", c->compute);
                    HIPCK(c, launch_tanh_forward(N, Fout, z->d, z->ld, h->d, h->ld, c->compute));
                    return DORY_OK;
                }
                return gemm(c, 0, 0, N, Fout, Fin, *ah, W, *z, h);
            }
            // last layer (CPU_comm.cpp:108-133)
            NEED(lab, layer, "lab");
            if ((rc = gemm(c, 0, 0, N, Fout, Fin, *ah, W, *z))) return rc;
            const uint32_t stt = (uint32_t)(N * 0.66);            // TRAIN_PORTION
            const uint32_t vend = stt + (uint32_t)(N * 0.1);      // VAL_PORTION
            c->val_rows = vend - stt;
            const float denom = (float)(c->globalV * 0.66);
            if ((rc = ensure_scratch(c, (size_t)(2 * ((vend - stt + 255) / 256) + 64) * sizeof(float)))) return rc;
            {
                Timed t(c, "loss", c->compute);
                // maskout copies (N - stt) floats starting at dense offset stt*cols (CPU_comm.cpp:464-471)
                HIPCK(c, launch_softmax_xent(N, Fout, z->d, z->ld, lab->d, lab->ld, g->d, g->ld, denom, stt,
                                             vend, (uint64_t)stt * Fout, (uint64_t)(N - stt), c->d_stat,
                                             c->scratch, c->compute));
            }
            if (layer > 0) {  // interGrad = d_output * W^T -> "grad"
                NEED(grad, layer, "grad");
                if ((rc = gemm(c, 0, 1, N, Fin, Fout, *g, W, *grad))) return rc;
            }
            return gemm(c, 1, 0, Fin, Fout, N, *ah, *g, dW);  // ah^T * d_output
        }
        // vtxNNBackwardGCN (CPU_comm.cpp:137-159)
        NEED(aTg, layer, "aTg");
        {
            Timed t(c, "loss", c->compute);
            HIPCK(c, launch_tanh_backward(N, Fout, aTg->d, aTg->ld, z->d, z->ld, g->d, g->ld, c->compute));
        }
        if (layer == 0 && tf_active(c)) return DORY_OK;   // dW0 follows in dory_aggregate(0, backward)
        if ((rc = gemm(c, 1, 0, Fin, Fout, N, *ah, *g, dW))) return rc;
        if (layer != 0) {
            NEED(grad, layer, "grad");
            return gemm(c, 0, 1, N, Fin, Fout, *g, W, *grad);
        }
        return DORY_OK;
    }
    if (c->gnn == DORY_GATMH) {  // extension: z = h*W ; backward dW = h^T dz, dh = dz W^T
        NEED(hh, layer, "h");
        NEED(z, layer, "z");
        const uint32_t zw = z->cols;
        if (dir == DORY_FORWARD) return gemm(c, 0, 0, N, zw, Fin, *hh, W, *z);
        NEED(dz, layer, "dz");
        if ((rc = gemm(c, 1, 0, Fin, zw, N, *hh, *dz, dW))) return rc;
        if (layer != 0) {
            NEED(dh, layer, "dh");
            return gemm(c, 0, 1, N, Fin, zw, *dz, W, *dh);
        }
        return DORY_OK;
    }
    // GAT
    Tensor *feats = layer == 0 ? find(c, 0, "h") : find(c, layer - 1, "ah");
    if (!feats) return fail(c, DORY_ERR_ARG, "apply_vertex GAT: input missing");
    if (dir == DORY_FORWARD) {  // vtxNNForwardGAT (CPU_comm.cpp:161-169)
        NEED(z, layer, "z");
        return gemm(c, 0, 0, N, Fout, Fin, *feats, W, *z);
    }
    // vtxNNBackwardGAT (CPU_comm.cpp:171-188)
    NEED(aTg, layer, "aTg");
    if ((rc = gemm(c, 1, 0, Fin, Fout, N, *feats, *aTg, dW))) return rc;
    if (layer != 0) {
        NEED(grad, layer - 1, "grad");
        return gemm(c, 0, 1, N, Fin, Fout, *aTg, W, *grad);
    }
    return DORY_OK;
}

int dory_apply_edge(dory_ctx *c, uint32_t layer, int dir) {
    CHECK_CTX(c);
    { int wrc = wait_halo(c); if (wrc) return wrc; }
    if (!c->prealloc || c->gnn == DORY_GCN) {
        if (c->prealloc && c->gnn == DORY_GCN) return DORY_OK;  // applyEdgeGCN is a no-op (gcn_ops.cpp:364-366)
        return fail(c, DORY_ERR_ARG, "apply_edge: preallocate first");
    }
    if (layer == 0 || layer > c->L) return fail(c, DORY_ERR_ARG, "apply_edge: layer %u out of range", layer);
    if (c->gnn == DORY_GATMH) {  // extension: attention scores per vertex and head; backward lives in aggregate
        if (dir != DORY_FORWARD) return DORY_OK;
        const uint32_t l0 = layer - 1, K = c->heads[l0];
        NEED(z, l0, "z"); NEED(el, l0, "el"); NEED(er, l0, "er");
        Timed t(c, "edge", c->compute);
        HIPCK(c, launch_gatmh_scores(c->N, K, z->cols / K, z->d, z->ld, c->weights[l0]["a_l"].d, c->weights[l0]["a_r"].d,
                                     el->d, er->d, el->ld, c->compute));
        return DORY_OK;
    }
    const uint32_t fl = layer - 1;  // "layer--; // YIFAN: fix this" (CPU_comm.cpp:33)
    const uint32_t F = c->dims[fl + 1];
    Tensor &a = c->weights[fl]["a_i"];
    NEED(z, fl, "z");
    NEED(az, fl, "az");
    if (dir == DORY_FORWARD) {  // edgNNForwardGAT (CPU_comm.cpp:190-203)
        NEED(arow, fl, "arow");
        Timed t(c, "edge", c->compute);
        HIPCK(c, launch_edge_forward_gat(c->N, F, c->colPtr, z->d, z->ld, a.d, az->d, c->cscVal, arow->d, c->compute));
        for (auto &f : c->gat_arow_valid) f = 0;   // "A" now holds this layer's scores only
        c->gat_arow_valid[fl] = 1;
        return DORY_OK;
    }
    // edgNNBackwardGAT (CPU_comm.cpp:205-242)
    NEED(grad, fl, "grad");
    NEED(dA, fl, "dA");
    NEED(cw, 0, "cw");
    NEED(drow, fl, "drow");
    Tensor &da = c->wgrads[fl]["a_i"];
    int rc = ensure_scratch(c, (size_t)(1024 * (size_t)F + F + c->N + 64) * sizeof(float));
    if (rc) return rc;
    float *r = c->scratch;            // F
    float *y = c->scratch + ((F + 63) & ~63u);   // N
    float *partial = y + ((c->N + 63) & ~63u);
    const size_t pbytes = c->scratch_bytes - (size_t)(partial - c->scratch) * sizeof(float);
    Timed t(c, "edge", c->compute);
    HIPCK(c, launch_edge_backward_gat(c->N, F, c->colPtr, grad->d, grad->ld, az->d, a.d, dA->d, cw->d, drow->d, c->compute));
    c->gat_drow_valid[fl] = 1;
    // r = grad^T cw ; da = z^T (z r)   [= (z^T z) r^T, CPU_comm.cpp:232-236, without the F x F matrix]
    HIPCK(c, launch_colsum_w(c->N, F, grad->d, grad->ld, cw->d, partial, pbytes, r, c->compute));
    HIPCK(c, launch_rowdot(c->N, F, z->d, z->ld, r, y, c->compute));
    HIPCK(c, launch_colsum_w(c->N, F, z->d, z->ld, y, partial, pbytes, da.d, c->compute));
    return DORY_OK;
}

int dory_predict_gat(dory_ctx *c, uint32_t layer) {
    CHECK_CTX(c);
    { int wrc = wait_halo(c); if (wrc) return wrc; }
    if (!c->prealloc || c->gnn == DORY_GCN || layer == 0 || layer > c->L)
        return fail(c, DORY_ERR_ARG, "predict_gat: bad state or layer");
    const uint32_t fl = layer - 1;
    if (c->gnn == DORY_GATMH) {
        NEED(lg, fl, "logits"); NEED(lab, fl, "lab"); NEED(gr, fl, "grad");
        Timed t(c, "loss", c->compute);
        HIPCK(c, launch_softmax_sub(c->N, lg->cols, lg->d, lg->ld, lab->d, lab->ld, gr->d, gr->ld, c->compute));
        return DORY_OK;
    }
    // Engine::predictGAT (gat_ops.cpp:246-265).  The reference reads the edge tensor
    // "az" where it means the aggregated "ah" (SURVEY.md 0-6); we use "ah".
    NEED(ah, fl, "ah");
    NEED(lab, fl, "lab");
    NEED(grad, fl, "grad");
    Timed t(c, "loss", c->compute);
    HIPCK(c, launch_softmax_sub(c->N, c->dims[layer], ah->d, ah->ld, lab->d, lab->ld, grad->d, grad->ld, c->compute));
    return DORY_OK;
}

int dory_train_stat(dory_ctx *c, float *acc_sum, float *loss_sum, uint32_t *val_rows) {
    CHECK_CTX(c);
    float h[2] = {0, 0};
    HIPCK(c, hipMemcpyAsync(h, c->d_stat, sizeof(h), hipMemcpyDeviceToHost, c->compute));
    HIPCK(c, hipStreamSynchronize(c->compute));
    if (acc_sum) *acc_sum = h[0];
    if (loss_sum) *loss_sum = h[1];
    if (val_rows) *val_rows = c->val_rows;
    return DORY_OK;
}

// ---------------------------------------------------------------------------------------
int dory_halo_plan(dory_ctx *c, int dir, const uint32_t *send_counts, const uint32_t *send_lvids,
                   const uint32_t *recv_counts, const uint32_t *recv_slots) {
    CHECK_CTX(c);
    if (!c->configured || !c->has_graph || (dir != 0 && dir != 1) || !send_counts || !recv_counts)
        return fail(c, DORY_ERR_ARG, "halo_plan: configure + graph_upload first / bad args");
    HaloPlan &p = c->plan[dir];
    const uint32_t P = c->numNodes;
    p.send_counts.assign(send_counts, send_counts + P);
    p.recv_counts.assign(recv_counts, recv_counts + P);
    p.send_off.assign(P + 1, 0);
    p.recv_off.assign(P + 1, 0);
    for (uint32_t i = 0; i < P; ++i) {
        p.send_off[i + 1] = p.send_off[i] + p.send_counts[i];
        p.recv_off[i + 1] = p.recv_off[i] + p.recv_counts[i];
    }
    p.send_total = p.send_off[P];
    p.recv_total = p.recv_off[P];
    const uint32_t G = dir == DORY_FORWARD ? c->Gsrc : c->Gdst;
    if (p.send_counts[c->nodeId] || p.recv_counts[c->nodeId]) return fail(c, DORY_ERR_ARG, "halo_plan: self entry must be empty");
    if (p.recv_total != G) return fail(c, DORY_ERR_ARG, "halo_plan: recv rows %u != ghost count %u", p.recv_total, G);
    for (uint32_t i = 0; i < p.send_total; ++i)
        if (send_lvids[i] >= c->N) return fail(c, DORY_ERR_ARG, "halo_plan: send lvid out of range");
    std::vector<char> seen(G, 0);
    for (uint32_t i = 0; i < p.recv_total; ++i) {
        if (recv_slots[i] >= G || seen[recv_slots[i]]) return fail(c, DORY_ERR_ARG, "halo_plan: recv slots must be a permutation of the ghost slots");
        seen[recv_slots[i]] = 1;
    }
    if (p.d_send_lvids) (void)hipFree(p.d_send_lvids);
    if (p.d_recv_slots) (void)hipFree(p.d_recv_slots);
    p.d_send_lvids = p.d_recv_slots = nullptr;
    int rc;
    if ((rc = upload_array(c, &p.d_send_lvids, send_lvids, p.send_total))) return rc;
    if ((rc = upload_array(c, &p.d_recv_slots, recv_slots, p.recv_total))) return rc;
    p.set = true;
    return DORY_OK;
}

int dory_comm_unique_id(void *id128) {
    if (!id128) return DORY_ERR_ARG;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return DORY_ERR_COMM;
    memcpy(id128, &id, sizeof(id));
    return DORY_OK;
}

int dory_comm_init(dory_ctx *c, const void *id128, int rank, int nranks) {
    CHECK_CTX(c);
    if (!id128 || rank < 0 || rank >= nranks) return fail(c, DORY_ERR_ARG, "comm_init: bad arguments");
    if (c->nccl) { ncclCommDestroy((ncclComm_t)c->nccl); c->nccl = nullptr; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm;
    NCCLCK(c, ncclCommInitRank(&comm, nranks, id, rank));
    c->nccl = comm;
    c->rank = rank;
    c->nranks = nranks;
    return DORY_OK;
}

// resolve (layer, dir) -> source tensor, ghost tensor, width, as Engine::scatterGCN/GAT do
static int halo_tensors(dory_ctx *c, uint32_t layer, int dir, Tensor **src, Tensor **ghost) {
    if (c->gnn == DORY_GCN && layer == 0 && dir == DORY_BACKWARD && tf_active(c)) {
        *src = find(c, 0, "g");      // transform-first: A^T g0 needs the ghost rows of g0
        *ghost = find(c, 0, "bgg");
    } else if (c->gnn == DORY_GCN) {
        if (layer == 0 || layer >= c->L) return fail(c, DORY_ERR_ARG, "halo: layer %u out of range", layer);
        if (dir == DORY_FORWARD) { *src = find(c, layer - 1, "h"); *ghost = find(c, layer, "fg"); }   // gcn_ops.cpp:205-214
        else { *src = find(c, layer, "grad"); *ghost = find(c, layer - 1, "bg"); }
    } else {
        if (layer == 0 || layer > c->L) return fail(c, DORY_ERR_ARG, "halo: layer %u out of range", layer);
        if (dir == DORY_FORWARD) { *src = find(c, layer - 1, "z"); *ghost = find(c, layer - 1, "fg_z"); }  // gat_ops.cpp:277-287
        else { *src = find(c, layer - 1, "grad"); *ghost = find(c, layer - 1, "bg_d"); }
    }
    if (!*src || !*ghost) return fail(c, DORY_ERR_ARG, "halo: tensors missing");
    return DORY_OK;
}

int dory_halo_pack(dory_ctx *c, uint32_t layer, int dir, float *send_buf) {
    CHECK_CTX(c);
    { int wrc = wait_halo(c); if (wrc) return wrc; }
    Tensor *src, *ghost;
    int rc = halo_tensors(c, layer, dir, &src, &ghost);
    if (rc) return rc;
    HaloPlan &p = c->plan[dir];
    if (!p.set) return fail(c, DORY_ERR_ARG, "halo_pack: no plan");
    Timed t(c, "halo", c->compute);
    HIPCK(c, launch_gather_rows(send_buf, src->d, src->ld, src->ld, p.d_send_lvids, p.send_total, c->compute));
    return DORY_OK;
}

int dory_halo_unpack(dory_ctx *c, uint32_t layer, int dir, const float *recv_buf) {
    CHECK_CTX(c);
    { int wrc = wait_halo(c); if (wrc) return wrc; }
    Tensor *src, *ghost;
    int rc = halo_tensors(c, layer, dir, &src, &ghost);
    if (rc) return rc;
    HaloPlan &p = c->plan[dir];
    if (!p.set) return fail(c, DORY_ERR_ARG, "halo_unpack: no plan");
    Timed t(c, "halo", c->compute);
    HIPCK(c, launch_scatter_rows(ghost->d, recv_buf, ghost->ld, ghost->ld, p.d_recv_slots, p.recv_total, c->compute));
    return DORY_OK;
}

int dory_halo_exchange(dory_ctx *c, uint32_t layer, int dir) {
    CHECK_CTX(c);
    { int wrc = wait_halo(c); if (wrc) return wrc; }
    if (c->numNodes == 1) return DORY_OK;  // no ghosts
    if (c->gnn == DORY_GATMH) return fail(c, DORY_ERR_ARG, "multi-head GAT extension: single partition only");
    Tensor *src, *ghost;
    int rc = halo_tensors(c, layer, dir, &src, &ghost);
    if (rc) return rc;
    HaloPlan &p = c->plan[dir];
    if (!p.set) return fail(c, DORY_ERR_ARG, "halo_exchange: no plan");
    if (!c->nccl) return fail(c, DORY_ERR_COMM, "halo_exchange: dory_comm_init not called");
    if (c->nranks != (int)c->numNodes) return fail(c, DORY_ERR_COMM, "halo_exchange: communicator size != num_nodes");
    const uint32_t w = src->ld;  // padded row width travels (keeps 16-B lanes)
    const size_t sb = (size_t)p.send_total * w * sizeof(float), rb = (size_t)p.recv_total * w * sizeof(float);
    if (sb > c->send_cap) {
        HIPCK(c, hipDeviceSynchronize());
        if (c->send_buf) (void)hipFree(c->send_buf);
        HIPCK(c, hipMalloc((void **)&c->send_buf, sb));
        c->send_cap = sb;
    }
    if (rb > c->recv_cap) {
        HIPCK(c, hipDeviceSynchronize());
        if (c->recv_buf) (void)hipFree(c->recv_buf);
        HIPCK(c, hipMalloc((void **)&c->recv_buf, rb));
        c->recv_cap = rb;
    }
    // comm stream waits for the producer of `src` on the compute stream
    HIPCK(c, hipEventRecord(c->ev_a, c->compute));
    HIPCK(c, hipStreamWaitEvent(c->comm, c->ev_a, 0));
    {
        Timed t(c, "halo", c->comm);
        HIPCK(c, launch_gather_rows(c->send_buf, src->d, src->ld, w, p.d_send_lvids, p.send_total, c->comm));
        ncclComm_t comm = (ncclComm_t)c->nccl;
        NCCLCK(c, ncclGroupStart());
        for (uint32_t peer = 0; peer < c->numNodes; ++peer) {
            if (peer == c->nodeId) continue;
            if (p.send_counts[peer])
                NCCLCK(c, ncclSend(c->send_buf + (size_t)p.send_off[peer] * w, (size_t)p.send_counts[peer] * w,
                                   ncclFloat, (int)peer, comm, c->comm));
            if (p.recv_counts[peer])
                NCCLCK(c, ncclRecv(c->recv_buf + (size_t)p.recv_off[peer] * w, (size_t)p.recv_counts[peer] * w,
                                   ncclFloat, (int)peer, comm, c->comm));
        }
        NCCLCK(c, ncclGroupEnd());
        HIPCK(c, launch_scatter_rows(ghost->d, c->recv_buf, ghost->ld, w, p.d_recv_slots, p.recv_total, c->comm));
    }
    // consumers on the compute stream wait for the ghosts: at once, or (halo_overlap) when
    // the first of them needs the ghost rows -- see wait_halo()
    HIPCK(c, hipEventRecord(c->ev_b, c->comm));
    if (c->opt["halo_overlap"]) c->halo_pending = true;
    else HIPCK(c, hipStreamWaitEvent(c->compute, c->ev_b, 0));
    return DORY_OK;
}

// ---------------------------------------------------------------------------------------
int dory_adam_config(dory_ctx *c, float learning_rate) {
    CHECK_CTX(c);
    c->adam.lr = learning_rate;
    c->adam.epochs = 1;
    c->lr_table_left = 0;   // an epoch graph's step-size table is refilled on its next launch
    return DORY_OK;
}

int dory_weight_update(dory_ctx *c, uint32_t layer) {
    CHECK_CTX(c);
    { int wrc = wait_halo(c); if (wrc) return wrc; }
    if (!c->prealloc || layer >= c->L) return fail(c, DORY_ERR_ARG, "weight_update: bad state or layer");
    // AdamOptimizer::nextIteration (src/weight-server/AdamOptimizer.cpp:29-34)
    const float b1p = (float)std::pow((double)0.9f, (double)c->adam.epochs);
    const float b2p = (float)std::pow((double)0.999f, (double)c->adam.epochs);
    const float lr_t = (float)(c->adam.lr * (std::sqrt((double)(1 - b2p))) / (1 - b1p));
    for (auto &kv : c->weights[layer]) {
        const std::string &name = kv.first;
        // the reference only updates "w"; a_i updates are faked on the weight server
        // (src/weight-server/weightserver.cpp:112-116) -- keep a_i fixed as it does.
        if (name != "w" && c->gnn != DORY_GATMH) continue;   // the extension trains a_l / a_r too
        Tensor &w = kv.second;
        Tensor &g = c->wgrads[layer][name];
        const uint64_t n = (uint64_t)w.rows * w.ld;
        if (c->numNodes > 1) {
            if (!c->nccl) return fail(c, DORY_ERR_COMM, "weight_update: dory_comm_init not called");
            // sum of per-partition updates (WeightTensor::localUpdate/ghostUpdate,
            // src/weight-server/weighttensor.cpp:131-166) as one RCCL all-reduce
            Timed t(c, "allreduce", c->compute);
            NCCLCK(c, ncclAllReduce(g.d, g.d, n, ncclFloat, ncclSum, (ncclComm_t)c->nccl, c->compute));
        }
        Timed t(c, "adam", c->compute);
        if (c->capturing)   // replayed epochs: step size from the table dory_epoch_graph_launch fills
            HIPCK(c, launch_adam_table(w.d, g.d, c->adam_m[layer][name].d, c->adam_v[layer][name].d, n, c->d_lr_table,
                                       c->d_replay_idx, c->compute));
        else
            HIPCK(c, launch_adam(w.d, g.d, c->adam_m[layer][name].d, c->adam_v[layer][name].d, n, lr_t, c->compute));
    }
    if (layer == 0 && !c->capturing) {   // "if(layer == 0) nextIteration();" (AdamOptimizer.cpp:49-50)
        c->adam.epochs += 1;
        c->lr_table_left = 0;            // eager step: a recorded epoch's table no longer lines up
    }
    return DORY_OK;
}

// ---------------------------------------------------------------------------------------
int dory_ctx_describe(dory_ctx *c, int *gnn_type, uint32_t *num_layers, uint32_t *node_id, uint32_t *num_nodes,
                      uint32_t *local_vtx_cnt) {
    CHECK_CTX(c);
    if (!c->configured || !c->has_graph) return fail(c, DORY_ERR_ARG, "ctx_describe: configure and graph_upload first");
    if (gnn_type) *gnn_type = c->gnn;
    if (num_layers) *num_layers = c->L;
    if (node_id) *node_id = c->nodeId;
    if (num_nodes) *num_nodes = c->numNodes;
    if (local_vtx_cnt) *local_vtx_cnt = c->N;
    return DORY_OK;
}

int dory_timing_enable(dory_ctx *c, int on) {
    CHECK_CTX(c);
    drain_timing(c);
    c->timing = on != 0;
    return DORY_OK;
}
int dory_timing_get(dory_ctx *c, const char *family, double *total_ms, uint64_t *launches) {
    CHECK_CTX(c);
    if (!family) return DORY_ERR_ARG;
    drain_timing(c);
    auto it = c->times.find(family);
    if (total_ms) *total_ms = it == c->times.end() ? 0.0 : it->second.total_ms;
    if (launches) *launches = it == c->times.end() ? 0 : it->second.launches;
    return DORY_OK;
}
int dory_timing_reset(dory_ctx *c) {
    CHECK_CTX(c);
    drain_timing(c);
    c->times.clear();
    return DORY_OK;
}
// ---------------------------------------------------------------------------------------
// Epoch graph: one epoch of C-ABI calls recorded into a hipGraph and replayed, so that a
// launch-bound epoch (Cora-sized graphs: ~35 kernels of a few microseconds) costs one
// graph launch.  No reference counterpart; single partition only (the exchange is not
// recorded).  Everything an epoch allocates lazily must exist already: run one eager epoch
// first.  Per-epoch host scalars do not survive recording, so Adam's step size comes from a
// device table indexed by a replay counter that the graph's last node bumps.
static float adam_lr_t(const dory_ctx *c, unsigned epochs) {   // AdamOptimizer::nextIteration, as dory_weight_update
    const float b1p = (float)std::pow((double)0.9f, (double)epochs);
    const float b2p = (float)std::pow((double)0.999f, (double)epochs);
    return (float)(c->adam.lr * (std::sqrt((double)(1 - b2p))) / (1 - b1p));
}

static void epoch_graph_drop_locked(dory_ctx *c) {
    if (c->capturing) {   // abandon a recording in progress
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(c->compute, &g);
        if (g) (void)hipGraphDestroy(g);
        c->capturing = false;
    }
    if (c->epoch_exec) (void)hipGraphExecDestroy(c->epoch_exec);
    if (c->epoch_graph) (void)hipGraphDestroy(c->epoch_graph);
    c->epoch_exec = nullptr;
    c->epoch_graph = nullptr;
    c->lr_table_left = 0;
}

int dory_epoch_graph_drop(dory_ctx *c) {
    CHECK_CTX(c);
    epoch_graph_drop_locked(c);
    return DORY_OK;
}

int dory_epoch_graph_begin(dory_ctx *c) {
    CHECK_CTX(c);
    if (!c->prealloc) return fail(c, DORY_ERR_ARG, "epoch_graph_begin: preallocate first");
    if (c->numNodes > 1) return fail(c, DORY_ERR_ARG, "epoch graph: single partition only (the halo exchange is not recorded)");
    if (c->capturing) return fail(c, DORY_ERR_ARG, "epoch_graph_begin: already recording");
    epoch_graph_drop_locked(c);
    if (!c->d_replay_idx) HIPCK(c, hipMalloc((void **)&c->d_replay_idx, 256));
    if (!c->d_lr_table) {
        c->lr_table_cap = 1024;
        HIPCK(c, hipMalloc((void **)&c->d_lr_table, c->lr_table_cap * sizeof(float)));
    }
    HIPCK(c, hipStreamSynchronize(c->compute));
    HIPCK(c, hipStreamBeginCapture(c->compute, hipStreamCaptureModeThreadLocal));
    c->capturing = true;
    return DORY_OK;
}

int dory_epoch_graph_end(dory_ctx *c) {
    CHECK_CTX(c);
    if (!c->capturing) return fail(c, DORY_ERR_ARG, "epoch_graph_end: not recording");
    hipError_t e = launch_bump_counter(c->d_replay_idx, c->compute);
    hipGraph_t g = nullptr;
    hipError_t e2 = hipStreamEndCapture(c->compute, &g);
    c->capturing = false;
    if (e != hipSuccess || e2 != hipSuccess || !g) {
        if (g) (void)hipGraphDestroy(g);
        return fail(c, DORY_ERR_HIP, "epoch_graph_end: recording failed (%s)", hipGetErrorString(e != hipSuccess ? e : e2));
    }
    c->epoch_graph = g;
    e = hipGraphInstantiate(&c->epoch_exec, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        c->epoch_exec = nullptr;
        epoch_graph_drop_locked(c);
        return fail(c, DORY_ERR_HIP, "epoch_graph_end: hipGraphInstantiate failed (%s)", hipGetErrorString(e));
    }
    return DORY_OK;
}

int dory_epoch_graph_launch(dory_ctx *c, uint32_t epochs) {
    CHECK_CTX(c);
    if (!c->epoch_exec) return fail(c, DORY_ERR_ARG, "epoch_graph_launch: no recorded epoch");
    for (uint32_t i = 0; i < epochs; ++i) {
        if (c->lr_table_left == 0) {
            // step sizes of the next replays, a pure function of the iteration count: filled well
            // ahead so that the host copy + counter reset happen once per lr_table_cap epochs
            HIPCK(c, hipStreamSynchronize(c->compute));   // previous replays have read the old table
            c->lr_table_host.resize(c->lr_table_cap);
            for (uint32_t k = 0; k < c->lr_table_cap; ++k) c->lr_table_host[k] = adam_lr_t(c, c->adam.epochs + k);
            HIPCK(c, hipMemcpy(c->d_lr_table, c->lr_table_host.data(), c->lr_table_cap * sizeof(float), hipMemcpyHostToDevice));
            HIPCK(c, hipMemset(c->d_replay_idx, 0, sizeof(uint32_t)));
            c->lr_table_left = c->lr_table_cap;
        }
        HIPCK(c, hipGraphLaunch(c->epoch_exec, c->compute));
        c->lr_table_left -= 1;
        c->adam.epochs += 1;
    }
    return DORY_OK;
}

int dory_transform_first_active(dory_ctx *c) {
    if (!c) return 0;
    std::lock_guard<std::mutex> lock(c->mu);
    return c->configured && tf_active(c) ? 1 : 0;
}

int dory_get_option(dory_ctx *c, const char *key, int64_t *value) {
    CHECK_CTX(c);
    if (!key || !value || c->opt.find(key) == c->opt.end()) return fail(c, DORY_ERR_ARG, "unknown option '%s'", key ? key : "(null)");
    *value = c->opt[key];
    return DORY_OK;
}

int dory_set_option(dory_ctx *c, const char *key, int64_t value) {
    CHECK_CTX(c);
    if (!key || c->opt.find(key) == c->opt.end()) return fail(c, DORY_ERR_ARG, "unknown option '%s'", key ? key : "(null)");
    c->opt[key] = value;
    return DORY_OK;
}

}  // extern "C"
