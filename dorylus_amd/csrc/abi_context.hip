// abi_context.hip -- C-ABI, part 1 (include/dorylus_hip.h): context, device tensor table, graph / tensor /
// weight uploads, options and timing.  The stage dispatch is in abi_stages.hip, the exchange and the weight update
// in abi_comm.hip.  Reference paths are relative to src/graph-server/ unless they start with src/.
#include "abi_internal.hpp"

namespace dory {

static std::string g_create_err;

int fail(dory_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}

void drain_timing(dory_ctx *c) {
    const dory_ctx::Pending *halo = nullptr;   // the last deferred exchange seen: the next "spmm_beside_halo" ran beside it
    for (auto &p : c->pending) {
        (void)hipEventSynchronize(p.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            c->times[p.fam].total_ms += ms;
            c->times[p.fam].launches += 1;
        }
        if (p.fam == "halo_deferred") halo = &p;
        if (p.fam == "spmm_beside_halo" && halo) {
            // both intervals on the clock of the exchange's first event: [0, he] and [ss, se]
            float he = 0.f, ss = 0.f, se = 0.f;
            if (hipEventElapsedTime(&he, halo->a, halo->b) == hipSuccess && hipEventElapsedTime(&ss, halo->a, p.a) == hipSuccess &&
                hipEventElapsedTime(&se, halo->a, p.b) == hipSuccess) {
                const float hidden = std::max(0.f, std::min(he, se) - std::max(0.f, ss));
                c->times["halo_hidden"].total_ms += hidden;
                c->times["halo_hidden"].launches += 1;
            }
            halo = nullptr;
        }
    }
    for (auto &p : c->pending) c->ev_pool.push_back({p.a, p.b});
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

// rows in the order of their MEDIAN source id (option spmm_order = 3, round 6 experiment): waves that run at the same time then
// read source rows from the same neighbourhood of the [local ; ghost] row space -- if the graph has one.  Per-row results are
// bit-identical whatever the schedule.
std::vector<uint32_t> median_source_order(const uint64_t *ptr, const uint32_t *idx, uint32_t N) {
    std::vector<uint32_t> med(N), o(N), tmp;
    for (uint32_t v = 0; v < N; ++v) {
        const uint64_t e0 = ptr[v], e1 = ptr[v + 1];
        if (e0 == e1) { med[v] = 0xFFFFFFFFu; continue; }
        tmp.assign(idx + e0, idx + e1);
        std::nth_element(tmp.begin(), tmp.begin() + (tmp.size() / 2), tmp.end());
        med[v] = tmp[tmp.size() / 2];
    }
    std::iota(o.begin(), o.end(), 0u);
    std::stable_sort(o.begin(), o.end(), [&](uint32_t a, uint32_t b) { return med[a] < med[b]; });
    return o;
}

int gemm(dory_ctx *c, int ta, int tb, uint32_t M, uint32_t N, uint32_t K, const Tensor &A,
         const Tensor &B, Tensor &C, Tensor *C2) {
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
// Ghost rows of the last halo exchange land on the comm stream; with "halo_overlap" the
// compute stream is only made to wait for them (event ev_b) by the first consumer.
int wait_halo(dory_ctx *c) {
    if (c->halo_pending) {
        if (c->local_pending.on) {   // in-process device transport: the peers' rows, unpack, ev_b
            int rc = local_exchange_finish(c);
            if (rc) return rc;
        }
        HIPCK(c, hipStreamWaitEvent(c->compute, c->ev_b, 0));
        c->halo_pending = false;
    }
    return DORY_OK;
}

int gat_materialize(dory_ctx *c, uint32_t layer, int which) {
    if (c->gnn != DORY_GAT || !c->prealloc) return DORY_OK;
    if ((which & 1) && layer < c->gat_az_stale.size() && c->gat_az_stale[layer]) {
        Tensor *az = find(c, layer, "az"), *azrow = find(c, layer, "azrow");
        if (az && azrow) HIPCK(c, launch_expand_rows_to_edges(c->N, c->colPtr, azrow->d, az->d, c->compute));
        c->gat_az_stale[layer] = 0;
    }
    if ((which & 2) && c->gat_A_stale_layer >= 0) {
        Tensor *arow = find(c, (uint32_t)c->gat_A_stale_layer, "arow");
        if (arow) HIPCK(c, launch_expand_rows_to_edges(c->N, c->colPtr, arow->d, c->cscVal, c->compute));
        c->gat_A_stale_layer = -1;
    }
    if ((which & 4) && layer < c->gat_dA_stale.size() && c->gat_dA_stale[layer]) {
        Tensor *dA = find(c, layer, "dA"), *drow = find(c, layer, "drow");
        if (dA && drow) HIPCK(c, launch_expand_rows_to_edges(c->N, c->colPtr, drow->d, dA->d, c->compute));
        c->gat_dA_stale[layer] = 0;
    }
    return DORY_OK;
}
static int gat_edge_tensor_hook(dory_ctx *c, uint32_t layer, const char *name, bool overwrite) {
    if (c->gnn != DORY_GAT || !name) return DORY_OK;
    const int which = !strcmp(name, "az") ? 1 : !strcmp(name, "A") ? 2 : !strcmp(name, "dA") ? 4 : 0;
    if (!which) return DORY_OK;
    if (!overwrite) return gat_materialize(c, layer, which);
    // the caller's values replace the tensor: nothing of ours is pending any more, and the per-vertex copy no longer describes it
    if (which == 1 && layer < c->gat_az_stale.size()) { c->gat_az_stale[layer] = 0; c->gat_azrow_valid[layer] = 0; }
    if (which == 2) c->gat_A_stale_layer = -1;
    if (which == 4 && layer < c->gat_dA_stale.size()) c->gat_dA_stale[layer] = 0;
    return DORY_OK;
}

// Transform-first order for a GCN layer (opt-in, no reference counterpart): when a layer's input is wider than its
// output, z_l = A (in_l W_l) gathers d[l+1]-wide rows instead of the d[l]-wide rows of (A in_l) W_l -- 128 instead of
// 602 floats per edge on Reddit's layer 0, 41 instead of 128 on layer 1.  Backward: u_l = A^T g_l (d[l+1] wide),
// dW_l = in_l^T u_l, and the gradient handed down is u_l W_l^T (= A^T (g_l W_l^T), the reference's aTg).  "ah"@l is
// not produced for such a layer.  Option gcn_transform_first: 1 = layer 0 only, 2 = every layer that narrows.
bool tf_layer(dory_ctx *c, uint32_t layer) {
    const int64_t mode = c->opt["gcn_transform_first"];
    if (c->gnn != DORY_GCN || c->L < 2 || layer >= c->L || mode == 0 || c->opt["adjacency_values_asymmetric"] != 0) return false;
    if (mode == 1 && layer != 0) return false;
    return c->dims[layer] > c->dims[layer + 1];
}
bool tf_active(dory_ctx *c) { return tf_layer(c, 0); }

}  // namespace dory

using namespace dory;

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
        hipMalloc((void **)&c->d_stat, 2 * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&c->sweep_stat, SWEEP_STAT_WORDS * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void **)&c->d_stat3, 256) != hipSuccess) {
        delete c;
        return fail(nullptr, DORY_ERR_HIP, "stream/event creation failed");
    }
    (void)hipMemset(c->d_stat, 0, 2 * sizeof(float));
    (void)hipMemset(c->sweep_stat, 0, SWEEP_STAT_WORDS * sizeof(uint32_t));
    c->own_compute = c->own_comm = true;
    c->opt["spmm_variant"] = 2;      // 2: K1s register-accumulating sweep over the blocked adjacency, 1: K1b (partial rows), 0: K1 only
    c->opt["spmm_sweep_flags"] = 0;          // K1s: reserved for experiments (bit 1 is the library's own "second launch" mark)
    c->opt["spmm_sweep_rows"] = 0;           // K1s: rows per lane group, 0 = by fill (2/4/6/8/10; tests and experiments)
    c->opt["spmm_sweep_pair"] = -1;          // K1s: two rows of a lane group as one stream of entries: -1 = launches of >= 3 slabs, 0 = never, 1 = always
    c->opt["spmm_sweep_loader"] = 1;         // K1s, 32-lane launches: wave 0 of a workgroup copies the next step's entries and offsets into LDS for all sixteen
    c->opt["spmm_sweep_loader_relief"] = 3;  // ... and the layout gives each of its two lane groups this many rows fewer per sweep (set before the layout is built)
    c->opt["spmm_sweep_reserve_cus"] = 4;    // K1s under an exchange in flight: CUs per XCD its sweeps leave to the RCCL kernels
    c->opt["spmm_sweep_layout"] = 3;         // K1s layout: 1 = spread the source rows over the blocks at random, 2 = deal the rows by degree (0: K1b's order -- graphs without structure only)
    c->opt["spmm_sweep_window_kb"] = 0;      // K1s: source window per block; 0 = 2432 KB (two live windows in one XCD's 4 MB L2), 3584 KB for partitions of <= 4 rows per lane group
    c->cus_per_xcd = (uint32_t)std::max(1, prop.multiProcessorCount / 8);
    {   // K1s's placement assumption (ctx.hpp): workgroup id & 7 = XCD, eight XCDs
        const uint32_t PG = 2048;
        uint32_t *dx = nullptr;
        std::vector<uint32_t> hx(PG, 0xFFu);
        if (hipMalloc((void **)&dx, PG * sizeof(uint32_t)) == hipSuccess) {
            bool ran = launch_xcd_probe(dx, PG, c->compute) == hipSuccess && hipStreamSynchronize(c->compute) == hipSuccess &&
                       hipMemcpy(hx.data(), dx, PG * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess;
            (void)hipFree(dx);
            // what K1s needs: workgroups with the same id & 7 share an XCD, the eight residues sit on eight different XCDs
            // (the dispatcher's round robin may start anywhere: the residue -> XCD map is a rotation, not the identity)
            uint32_t seen = 0, match = 0;
            for (uint32_t i = 0; i < PG; ++i) { if (hx[i] < 16) seen |= 1u << hx[i]; match += hx[i] == hx[i & 7u]; }
            uint32_t firsts = 0;
            for (uint32_t r = 0; r < 8; ++r) if (hx[r] < 16) firsts |= 1u << hx[r];
            c->xcd_count = (uint32_t)__builtin_popcount(seen);
            c->xcd_mapping_ok = ran && match == PG && c->xcd_count == 8 && __builtin_popcount(firsts) == 8;
        } else {
            c->xcd_mapping_ok = false;
        }
    }
    c->opt["spmm_xcd_assume_mismatch"] = 0;  // testing: treat the placement check as failed (the gated / ungated choice is then made by measurement)
    c->opt["spmm_slab"] = 0;
    c->opt["spmm_order"] = 1;    // K1: rows longest first -- 1 = when the degrees are skewed (max > 8 x mean), 2 = always, 0 = never;
                                 // 3 (before dory_graph_upload) = rows by their median source id instead (experiment, profiles/HISTORY.md)
    c->opt["spmm_blk_group"] = 32;   // K1b: lanes per row (slab = 4*group floats = 512 B)
    c->opt["spmm_blk_force_split"] = 0;   // testing: always launch local / ghost source blocks separately
    c->opt["halo_overlap"] = 1;      // let local-source blocks of the next SpMM run under the exchange
    c->opt["gat_lazy_edge_tensors"] = 1;  // GAT prototype: az / A / dA (one value per destination) are written per edge only when read (download, raw pointer, K1's per-edge path)
    c->opt["gat_reuse_nsum"] = 1;         // GAT prototype: the backward's dA-weighted aggregation from the forward's neighbour sum (abi_stages.hip)
    c->opt["spmm_edge_split"] = 1;        // K1 on GCN partitions with ghosts: every row's local-source edges first (set before dory_graph_upload)
    c->opt["spmm_sweep_cus"] = 0;         // K1s / GAT sweeps: workgroups per sweep and XCD (0 = all CUs of an XCD); see dory_set_option
    c->opt["local_timeout_ms"] = 30000;   // in-process device transport: how long a rank's host thread waits for a peer's host thread
    c->opt["adjacency_values_asymmetric"] = 0;   // set by dory_partition_upload for undirected / unknown builds: csrVal != cscVal^T
    c->opt["gatmh_bwd_phase"] = 0;       // multi-head GAT backward: 0 = whole sweep (exchanging the ghost rows itself), 1 / 2 = first / second phase only (callers with their own transport)
    c->opt["gatmh_blocked"] = 1;         // multi-head GAT: source-blocked (L2-resident) gathers where the blocked adjacency applies
    c->opt["gatmh_el_on_the_fly"] = 1;       // multi-head GAT, blocked forward with fused statistics, heads of <= 16 features: el[src] from the gathered row instead of a second gather
    c->opt["gatmh_sweep"] = 1;               // multi-head GAT: the edge passes on K1s's skeleton (gat_mh_sweep.hip) where the sweep layout and the shape apply (1: forward)
    c->opt["gatmh_src_window_kb"] = 0;       // multi-head GAT: source window of the OUT-edge sweep layout in KB of 512-byte rows (0 = as the forward's, 4608)
    c->opt["gatmh_sweep_rows"] = 0;          // rows per lane group of the multi-head GAT contexts' sweep layouts (0 = by fill, at most 8)
    c->opt["gatmh_fused_stats"] = 1;         // multi-head GAT, blocked forward: online softmax per source block + merge in the reduce (0: separate statistics pass first)
    c->opt["gcn_cache_ah0"] = 0;         // GCN: keep ah@0 = A_hat x across epochs while x, fg@0 and the adjacency are unchanged (opt-in; the reference recomputes it)
    c->opt["gcn_transform_first"] = 0;   // GCN layers as A(XW) instead of (AX)W where the input is wider than the output: 1 = layer 0, 2 = all (see tf_layer)
    c->opt["epoch_graph"] = 0;       // engine: replay a recorded epoch (hipGraph) when the partition is alone
    c->opt["spmm_blk_nb"] = 0;       // K1b: number of source blocks (0 = auto, ~3.75 MB windows)
    *out = c;
    return DORY_OK;
}

static void free_graph(dory_ctx *c) {
    void *ps[] = {c->colPtr, c->rowPtr, c->rowIdx, c->colIdx, c->cscVal, c->csrVal, c->norm, c->orderIn, c->orderOut,
                  c->splitIn, c->splitOut};
    for (void *p : ps)
        if (p) (void)hipFree(p);
    c->colPtr = c->rowPtr = nullptr;
    c->rowIdx = c->colIdx = nullptr;
    c->cscVal = c->csrVal = c->norm = nullptr;
    c->orderIn = c->orderOut = nullptr;
    c->splitIn = c->splitOut = nullptr;
    c->nIntIn = c->nIntOut = 0;
    for (EdgeSplit *E : {&c->esIn, &c->esOut}) {
        if (E->idx) (void)hipFree(E->idx);
        if (E->val) (void)hipFree(E->val);
        if (E->mid) (void)hipFree(E->mid);
        *E = EdgeSplit{};
    }
    for (LongRowsDev *L : {&c->longIn, &c->longOut}) {
        if (L->rows) (void)hipFree(L->rows);
        if (L->row_chunk_ptr) (void)hipFree(L->row_chunk_ptr);
        if (L->chunks) (void)hipFree(L->chunks);
        *L = LongRowsDev{};
    }
    free_blocked(&c->blkIn);
    free_blocked(&c->blkOut);
    free_blocked(&c->blkIn16);
    free_blocked(&c->blkOut16);
    c->blkIn16_built = c->blkOut16_built = false;
    free_blocked(&c->swpIn);
    free_blocked(&c->swpOut);
    c->swpIn_built = c->swpOut_built = c->swpIn_na = c->swpOut_na = false;
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
    if (c->local) {   // leave the group: peers find a null entry, not a dangling pointer
        std::lock_guard<std::mutex> lk(c->local->mu);
        for (auto &q : c->local->ctx) if (q == c) q = nullptr;
    }
    for (hipEvent_t e : {c->ev_sent[0], c->ev_sent[1], c->ev_cons[0], c->ev_cons[1], c->ev_gready[0], c->ev_gready[1], c->ev_gdone[0], c->ev_gdone[1]})
        if (e) (void)hipEventDestroy(e);
    if (c->ar_tmp) (void)hipFree(c->ar_tmp);
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
    if (c->sweep_stat) (void)hipFree(c->sweep_stat);
    if (c->d_stat3) (void)hipFree(c->d_stat3);
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
    { int wrc = wait_halo(c); if (wrc) return wrc; }   // (a deferred exchange nobody consumed yet: its ghosts land first)
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
    c->ah0_valid = false;
    if (!column_ptrs || !row_ptrs || (N && !vtx_norms) || (nnz_in && (!row_idxs || !csc_values)) ||
        (nnz_out && (!column_idxs || !csr_values)))
        return fail(c, DORY_ERR_ARG, "dory_graph_upload: null array");
    epoch_graph_drop_locked(c);   // a recorded epoch points at the adjacency (and its blocked copies) freed below
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
    const bool by_median = c->opt["spmm_order"] == 3;    // (set before the upload: the schedule is built here)
    auto oi = by_median ? median_source_order(column_ptrs, row_idxs, N) : degree_order(column_ptrs, N);
    auto oo = by_median ? median_source_order(row_ptrs, column_idxs, N) : degree_order(row_ptrs, N);
    {   // is the degree distribution skewed enough for the longest-first row schedule to pay?  (Amazon-size uniform graph:
        // 129 ms per epoch in row order against 134 longest first; R-MAT of the same size: 118 against 104)
        auto skew = [&](const uint64_t *ptr, uint64_t nnz) {
            uint64_t mx = 0;
            for (uint32_t v = 0; v < N; ++v) mx = std::max<uint64_t>(mx, ptr[v + 1] - ptr[v]);
            return N > 0 && mx * (uint64_t)N > 8ull * nnz + 8ull * N;
        };
        c->skewIn = skew(column_ptrs, nnz_in);
        c->skewOut = skew(row_ptrs, nnz_out);
    }
    if ((rc = upload_array(c, &c->orderIn, oi.data(), (uint64_t)N))) return rc;
    if ((rc = upload_array(c, &c->orderOut, oo.data(), (uint64_t)N))) return rc;
    for (int d = 0; d < 2; ++d) {   // K1's interior / boundary row split (partitions with ghosts), each direction on its own
        if ((d == 0 ? Gsrc : Gdst) == 0) continue;
        const uint64_t *ptr = d == 0 ? column_ptrs : row_ptrs;
        const uint32_t *idx = d == 0 ? row_idxs : column_idxs;
        const std::vector<uint32_t> &ord = d == 0 ? oi : oo;
        std::vector<uint32_t> interior, boundary;
        for (uint32_t v : ord) {   // keeps the longest-row-first order inside both parts
            bool local = true;
            for (uint64_t e = ptr[v]; e < ptr[v + 1] && local; ++e) local = idx[e] < N;
            (local ? interior : boundary).push_back(v);
        }
        (d == 0 ? c->nIntIn : c->nIntOut) = (uint32_t)interior.size();
        interior.insert(interior.end(), boundary.begin(), boundary.end());
        if ((rc = upload_array(c, d == 0 ? &c->splitIn : &c->splitOut, interior.data(), (uint64_t)N))) return rc;
    }
    for (int d = 0; d < 2; ++d) {   // K1's hub rows (spmm.hip: long rows)
        LongRowsHost h;
        plan_long_rows(d == 0 ? column_ptrs : row_ptrs, N, &h);
        LongRowsDev &L = d == 0 ? c->longIn : c->longOut;
        L.nrows = (uint32_t)h.rows.size();
        L.nchunks = (uint32_t)(h.chunks.size() / 6);
        if (!L.nchunks) continue;
        if ((rc = upload_array(c, &L.rows, h.rows.data(), h.rows.size()))) return rc;
        if ((rc = upload_array(c, &L.row_chunk_ptr, h.row_chunk_ptr.data(), h.row_chunk_ptr.size()))) return rc;
        if ((rc = upload_array(c, &L.chunks, h.chunks.data(), h.chunks.size()))) return rc;
    }
    // K1's local-first edge order (ctx.hpp: EdgeSplit) for GCN partitions with ghosts and without hub rows
    for (int d = 0; d < 2 && c->gnn == DORY_GCN && c->opt["spmm_edge_split"]; ++d) {
        if ((d == 0 ? Gsrc : Gdst) == 0 || (d == 0 ? c->longIn : c->longOut).nchunks) continue;
        const uint64_t *ptr = d == 0 ? column_ptrs : row_ptrs;
        const uint32_t *idx = d == 0 ? row_idxs : column_idxs;
        const float *val = d == 0 ? csc_values : csr_values;
        const uint64_t nnz = d == 0 ? nnz_in : nnz_out;
        std::vector<uint32_t> i2(nnz);
        std::vector<float> v2(nnz);
        std::vector<uint64_t> mid(N);
#pragma omp parallel for schedule(dynamic, 4096)
        for (int64_t v = 0; v < (int64_t)N; ++v) {
            uint64_t w = ptr[v];
            for (uint64_t e = ptr[v]; e < ptr[v + 1]; ++e)
                if (idx[e] < N) { i2[w] = idx[e]; v2[w] = val[e]; ++w; }
            mid[v] = w;
            for (uint64_t e = ptr[v]; e < ptr[v + 1]; ++e)
                if (idx[e] >= N) { i2[w] = idx[e]; v2[w] = val[e]; ++w; }
        }
        EdgeSplit &E = d == 0 ? c->esIn : c->esOut;
        if ((rc = upload_array(c, &E.idx, i2.data(), nnz))) return rc;
        if ((rc = upload_array(c, &E.val, v2.data(), nnz))) return rc;
        if ((rc = upload_array(c, &E.mid, mid.data(), (uint64_t)N))) return rc;
    }
    c->has_graph = true;
    return DORY_OK;
}

int dory_preallocate(dory_ctx *c) {
    CHECK_CTX(c);
    c->ah0_valid = false;
    if (!c->configured || !c->has_graph) return fail(c, DORY_ERR_ARG, "dory_preallocate: configure and graph_upload first");
    epoch_graph_drop_locked(c);   // a recorded epoch points at the tensors freed below
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
        for (uint32_t l = 0; l < L && L >= 2; ++l) {   // transform-first order (option gcn_transform_first)
            if (d[l] <= d[l + 1]) continue;
            mk(l, "xw", N, d[l + 1]);          // in_l W_l
            mk(l, "fgxw", c->Gsrc, d[l + 1]);  // its ghost rows (layer 0: transformed locally from fg@0, else exchanged)
            mk(l, "u", N, d[l + 1]);           // A^T g_l
            mk(l, "bgg", c->Gdst, d[l + 1]);   // ghost rows of g_l (backward exchange)
        }
    } else if (c->gnn == DORY_GATMH) {  // extension (no reference counterpart): see dory_gatmh_heads
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
            mk(l, "st", N, 4 * K);               // (er, m, 1/den, t) per (v,k): what the source-side sweep gathers per edge
            mk(l, "op", N, zw);                  // the part of "o" that came over edges on LeakyReLU's positive branch (sweep forward)
            mk(l, "dpos", N, K);                 // and the attention mass of those edges
            // partitioned runs: ghost sources of the in-edges (z exchanged forward, el/er recomputed from it) and
            // ghost destinations of the out-edges (dO and st exchanged between the two phases of the backward sweep)
            mk(l, "fg_z", c->Gsrc, zw);
            mk(l, "fg_el", c->Gsrc, K);
            mk(l, "fg_er", c->Gsrc, K);
            mk(l, "bg_do", c->Gdst, zw);
            mk(l, "bg_st", c->Gdst, 4 * K);
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
        for (uint32_t l = 0; l < L; ++l) mk(l, "azrow", N, 1);  // per-destination value of "az"
        c->gat_az_stale.assign(L, 0);
        c->gat_dA_stale.assign(L, 0);
        c->gat_azrow_valid.assign(L, 0);
        c->gat_A_stale_layer = -1;
        for (uint32_t l = 0; l < L; ++l) mk(l, "nsum", N, d[l + 1]);   // unweighted neighbour sum of z (kept from the forward for the backward aggregation)
        mk(0, "ones", N, 1);
        c->gat_arow_valid.assign(L, 0);
        c->gat_drow_valid.assign(L, 0);
        c->gat_nsum_valid.assign(L, 0);
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
    if (c->opt["spmm_variant"] == 2 && N > 0 && c->gnn == DORY_GATMH && c->opt["gatmh_sweep"]) {
        // the sweep layouts (K1s's even layout; the deal is made for the 32-lane launches) and the gate counters now
        uint32_t maxld = 0;
        for (uint32_t l = 0; l < L; ++l) maxld = std::max(maxld, pad_ld(l == L - 1 ? d[l + 1] * c->heads[l] : d[l + 1]));
        const int group = blk_group_for(c, maxld);
        if ((rc = ensure_sweep(c, true, group))) return rc;
        if ((rc = ensure_sweep(c, false, group))) return rc;
        c->gatmh_fwd_swept.assign(L, 0);
        size_t need = 0;
        for (const BlockedAdj *S : {&c->swpIn, &c->swpOut})
            if (S->nb)
                for (uint32_t l = 0; l < L; ++l) {   // every layer's own launch shape: its leading dimension (a 96-float layer runs 16-lane
                    // groups on two slabs) and the group blk_group_for() gives it; launches walk fewer rows per group than the
                    // layout deals (more sweeps, more counters): 2 is the least
                    const uint32_t ld_l = pad_ld(l == L - 1 ? d[l + 1] * c->heads[l] : d[l + 1]);
                    need = std::max(need, sweep_scratch_bytes(*S, ld_l, blk_group_for(c, ld_l), std::min<uint32_t>(32u, c->cus_per_xcd), S->nb, 2));
                }
        if (need > c->partial_bytes) {
            if (c->partial) (void)hipFree(c->partial);
            c->partial = nullptr;
            c->partial_bytes = 0;
            HIPCK(c, hipMalloc((void **)&c->partial, need));
            c->partial_bytes = need;
        }
    }
    if (c->opt["spmm_variant"] >= 1 && N > 0 && c->gnn == DORY_GATMH && c->opt["gatmh_blocked"]) {
        // the extension's forward sum gathers through the same source-blocked copy of the in-edges
        uint32_t maxld = 0;
        for (uint32_t l = 0; l < L; ++l) maxld = std::max(maxld, pad_ld(l == L - 1 ? d[l + 1] * c->heads[l] : d[l + 1]));
        if ((rc = ensure_blocked(c, true, blk_group_for(c, maxld)))) return rc;
        if ((rc = ensure_blocked(c, false, blk_group_for(c, maxld)))) return rc;   // backward, source side
        uint32_t minld = maxld;
        for (uint32_t l = 0; l < L; ++l) minld = std::min(minld, pad_ld(l == L - 1 ? d[l + 1] * c->heads[l] : d[l + 1]));
        if (maxld >= 128 && minld < 128 && !c->blkIn_na && !c->blkOut_na) {   // narrow layers beside wide ones: their own pair (256-B slabs)
            if ((rc = ensure_blocked(c, true, 16, true))) return rc;
            if ((rc = ensure_blocked(c, false, 16, true))) return rc;
        }
        const uint32_t nbmax = std::max(c->blkIn.nb, c->blkOut.nb);
        size_t need = (size_t)nbmax * N * (maxld + 64) * sizeof(float);            // + per-(block,row,head) partials
        need = std::max(need, (size_t)std::max(c->blkIn16.nb, c->blkOut16.nb) * N * (std::min<uint32_t>(maxld, 96u) + 64) * sizeof(float));
        if (nbmax && need <= ((size_t)48 << 30) && need > c->partial_bytes) {
            if (c->partial) (void)hipFree(c->partial);
            c->partial = nullptr;
            c->partial_bytes = 0;
            HIPCK(c, hipMalloc((void **)&c->partial, need));
            c->partial_bytes = need;
        }
    }
    if (c->opt["spmm_variant"] >= 1 && N > 0 && c->gnn != DORY_GATMH) {   // K1b: regroup the edges now, not inside the first epoch
        uint32_t minld = 0xFFFFFFFFu;
        for (uint32_t l = 0; l < L; ++l) {
            const uint32_t w = c->gnn == DORY_GCN ? (l == 0 ? d[0] : d[l]) : d[l + 1];
            minld = std::min(minld, pad_ld(w));
        }
        uint32_t maxld = 0;
        for (uint32_t l = 0; l <= L; ++l) maxld = std::max(maxld, pad_ld(d[l]));
        const int group = blk_group_for(c, maxld);   // block size for the widest rows (most of the traffic)
        if (minld >= 32 && c->opt["spmm_variant"] == 2) {
            // K1s: its layouts and the gate counters of the largest launch now, so that nothing is built or allocated
            // inside an epoch (a partition it does not take -- too small, too large -- keeps K1 / builds K1b on demand)
            if ((rc = ensure_sweep(c, true, group))) return rc;
            if ((rc = ensure_sweep(c, false, group))) return rc;
            size_t need = 0;
            for (const BlockedAdj *S : {&c->swpIn, &c->swpOut})
                if (S->nb) need = std::max(need, sweep_scratch_bytes(*S, maxld, group, std::min<uint32_t>(32u, c->cus_per_xcd), S->nb, (int)c->opt["spmm_sweep_rows"]));
            if (need > c->partial_bytes) {
                if (c->partial) (void)hipFree(c->partial);
                c->partial = nullptr;
                c->partial_bytes = 0;
                HIPCK(c, hipMalloc((void **)&c->partial, need));
                c->partial_bytes = need;
            }
            // the slots of the split rows' pieces too (skewed graphs): an epoch recorded into a hipGraph right after a
            // re-upload must not find anything left to allocate
            const uint32_t nslots = std::max(c->swpIn.nslots, c->swpOut.nslots);
            if (nslots && (rc = ensure_scratch(c, (size_t)nslots * maxld * sizeof(float)))) return rc;
        } else if (minld >= 32) {
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
    c->gat_ones_set = false;
    c->prealloc = true;
    return DORY_OK;
}

// ---------------------------------------------------------------------------------------
int dory_tensor_info(dory_ctx *c, uint32_t layer, const char *name, uint64_t *rows, uint32_t *cols,
                     uint32_t *ld, void **device_ptr) {
    CHECK_CTX(c);
    Tensor *t = name ? find(c, layer, name) : nullptr;
    if (!t) return fail(c, DORY_ERR_ARG, "no tensor '%s' at layer %u", name ? name : "(null)", layer);
    if (device_ptr) { int hrc = gat_edge_tensor_hook(c, layer, name, false); if (hrc) return hrc; }   // a raw pointer: the data behind it must be current
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
    if (layer == 0) c->ah0_valid = false;                                    // x / fg@0 / ah@0 may have changed
    if (!strcmp(name, "A")) for (auto &f : c->gat_arow_valid) f = 0;          // caller-supplied edge weights: general path
    if (!strcmp(name, "dA") && layer < c->gat_drow_valid.size()) c->gat_drow_valid[layer] = 0;
    if ((!strcmp(name, "z") || !strcmp(name, "fg_z")) && layer < c->gat_nsum_valid.size()) c->gat_nsum_valid[layer] = 0;
    { int hrc = gat_edge_tensor_hook(c, layer, name, true); if (hrc) return hrc; }
    return upload_dense(c, *t, host);
}

int dory_tensor_download(dory_ctx *c, uint32_t layer, const char *name, float *host) {
    CHECK_CTX(c);
    { int wrc = wait_halo(c); if (wrc) return wrc; }
    Tensor *t = name ? find(c, layer, name) : nullptr;
    if (!t || !host) return fail(c, DORY_ERR_ARG, "tensor_download: no tensor '%s' at layer %u", name ? name : "(null)", layer);
    { int hrc = gat_edge_tensor_hook(c, layer, name, false); if (hrc) return hrc; }
    return download_dense(c, *t, host);
}

int dory_tensor_fill_uniform(dory_ctx *c, uint32_t layer, const char *name, uint64_t seed, float lo,
                             float hi, const uint32_t *global_row_ids) {
    CHECK_CTX(c);
    { int wrc = wait_halo(c); if (wrc) return wrc; }
    Tensor *t = name ? find(c, layer, name) : nullptr;
    if (!t) return fail(c, DORY_ERR_ARG, "tensor_fill: no tensor '%s' at layer %u", name ? name : "(null)", layer);
    if (layer == 0) c->ah0_valid = false;
    if ((!strcmp(name, "z") || !strcmp(name, "fg_z")) && layer < c->gat_nsum_valid.size()) c->gat_nsum_valid[layer] = 0;
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

int dory_weight_grad_set(dory_ctx *c, uint32_t layer, const char *name, const float *host) {
    CHECK_CTX(c);
    Tensor *t = name ? findw(c->wgrads, layer, name) : nullptr;
    if (!t || !host) return fail(c, DORY_ERR_ARG, "weight_grad_set: no gradient '%s' at layer %u", name ? name : "(null)", layer);
    return upload_dense(c, *t, host);
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
    if (!strcmp(family, "spmm_gate_timeouts")) {   // not a kernel family: launches = gate timeouts, total_ms = ungated launches
        uint32_t st[4] = {0, 0, 0, 0};
        if (c->capturing) return fail(c, DORY_ERR_ARG, "spmm_gate_timeouts: not while an epoch graph is being recorded (reading synchronises the stream)");
        HIPCK(c, hipStreamSynchronize(c->compute));
        HIPCK(c, hipMemcpy(st, c->sweep_stat, sizeof(st), hipMemcpyDeviceToHost));
        if (total_ms) *total_ms = (double)st[2];
        if (launches) *launches = st[0];
        return DORY_OK;
    }
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
int dory_transform_first_active(dory_ctx *c) {
    if (!c) return 0;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!c->configured) return 0;
    for (uint32_t l = 0; l < c->L; ++l)
        if (tf_layer(c, l)) return 1;
    return 0;
}

int dory_transform_first_layer(dory_ctx *c, uint32_t layer) {
    if (!c) return 0;
    std::lock_guard<std::mutex> lock(c->mu);
    return c->configured && tf_layer(c, layer) ? 1 : 0;
}

int dory_get_option(dory_ctx *c, const char *key, int64_t *value) {
    CHECK_CTX(c);
    if (key && value && !strcmp(key, "gcn_cache_ah0_skips")) {   // read-only: layer-0 aggregations answered from the cached ah@0
        *value = (int64_t)c->ah0_skips;
        return DORY_OK;
    }
    if (key && value && !strcmp(key, "epoch_graph_recorded")) {   // read-only: does the ctx still hold a recorded epoch?
        *value = c->epoch_exec ? 1 : 0;
        return DORY_OK;
    }
    // read-only counters of the K1s gates (device words that outlive the launches): timeouts = a sweep's workgroups were
    // not co-resident within the polling bound; ungated launches = launches that ran without gates while the context
    // backed off after a timeout (same results, unsynchronised rate)
    if (key && value && !strcmp(key, "spmm_gates_rearm")) {   // (write-only action; reads as "is a back-off pending": launch number below a horizon)
        uint32_t st[SWEEP_STAT_WORDS] = {0};
        if (c->capturing) return fail(c, DORY_ERR_ARG, "spmm_gates_rearm: not while an epoch graph is being recorded (reading synchronises the stream)");
        HIPCK(c, hipStreamSynchronize(c->compute));
        HIPCK(c, hipMemcpy(st, c->sweep_stat, sizeof(st), hipMemcpyDeviceToHost));
        *value = (st[4] < st[1] || st[4] < st[3]) ? 1 : 0;
        return DORY_OK;
    }
    if (key && value && !strcmp(key, "spmm_xcd_mapping_ok")) { *value = c->xcd_mapping_ok ? 1 : 0; return DORY_OK; }   // read-only: dory_create's check
    if (key && value && !strcmp(key, "spmm_xcd_count")) { *value = c->xcd_count; return DORY_OK; }
    if (key && value && !strcmp(key, "spmm_xcd_policy")) { *value = c->xcd_policy; return DORY_OK; }                  // -1 undecided / not needed, 0 gated, 8 ungated
    if (key && value && !strcmp(key, "spmm_xcd_gated_us")) { *value = (int64_t)(c->xcd_gated_ms * 1e3f); return DORY_OK; }
    if (key && value && !strcmp(key, "spmm_xcd_ungated_us")) { *value = (int64_t)(c->xcd_ungated_ms * 1e3f); return DORY_OK; }
    if (key && value && (!strcmp(key, "spmm_gate_timeouts") || !strcmp(key, "spmm_ungated_launches"))) {
        uint32_t st[4] = {0, 0, 0, 0};
        if (c->capturing) return fail(c, DORY_ERR_ARG, "%s: not while an epoch graph is being recorded (reading synchronises the stream)", key);
        HIPCK(c, hipStreamSynchronize(c->compute));
        HIPCK(c, hipMemcpy(st, c->sweep_stat, sizeof(st), hipMemcpyDeviceToHost));
        *value = !strcmp(key, "spmm_gate_timeouts") ? st[0] : st[2];
        return DORY_OK;
    }
    if (!key || !value || c->opt.find(key) == c->opt.end()) return fail(c, DORY_ERR_ARG, "unknown option '%s'", key ? key : "(null)");
    *value = c->opt[key];
    return DORY_OK;
}

int dory_debug_occupy_cus(dory_ctx *c, uint32_t workgroups, uint64_t usec) {
    CHECK_CTX(c);
    if (usec > 2000000) return fail(c, DORY_ERR_ARG, "debug_occupy_cus: at most 2 s");
    HIPCK(c, launch_occupy_cus(workgroups, usec, c->comm));
    return DORY_OK;
}

int dory_set_option(dory_ctx *c, const char *key, int64_t value) {
    CHECK_CTX(c);
    if (key && !strcmp(key, "spmm_gates_rearm")) {   // write-only: end a K1s gate back-off now (both launch classes); the counters stay.
        // For a caller that has just changed what caused the timeouts (bench.py trying another spmm_sweep_reserve_cus).
        if (c->capturing) return fail(c, DORY_ERR_ARG, "spmm_gates_rearm: not while an epoch graph is being recorded (it synchronises the stream)");
        HIPCK(c, hipStreamSynchronize(c->compute));
        const uint32_t zero = 0;
        HIPCK(c, hipMemcpy(c->sweep_stat + 1, &zero, sizeof(zero), hipMemcpyHostToDevice));
        HIPCK(c, hipMemcpy(c->sweep_stat + 3, &zero, sizeof(zero), hipMemcpyHostToDevice));
        return DORY_OK;
    }
    if (key && !strcmp(key, "spmm_sweep_cus")) {   // workgroups per sweep and XCD of K1s and the GAT sweeps (0: every CU of an XCD).  For several
        // contexts that SHARE a device (dory_comm_init_local: P ranks on one GPU): each takes its share of the CUs, so that
        // the ranks' gated sweeps are co-resident beside each other.  Before dory_graph_upload (the layouts are dealt for it).
        if (c->has_graph) return fail(c, DORY_ERR_ARG, "spmm_sweep_cus: set it before the graph is uploaded");
        hipDeviceProp_t prop;
        HIPCK(c, hipGetDeviceProperties(&prop, c->device));
        const uint32_t all = (uint32_t)std::max(1, prop.multiProcessorCount / 8);
        if (value < 0 || value > (int64_t)all) return fail(c, DORY_ERR_ARG, "spmm_sweep_cus: 0..%u", all);
        c->cus_per_xcd = value == 0 ? all : (uint32_t)value;
        c->opt[key] = value;
        return DORY_OK;
    }
    if (!key || c->opt.find(key) == c->opt.end()) return fail(c, DORY_ERR_ARG, "unknown option '%s'", key ? key : "(null)");
    c->opt[key] = value;
    c->ah0_valid = false;   // (another kernel variant sums in another order: a cached ah@0 is only kept across identical settings)
    return DORY_OK;
}

}  // extern "C"
