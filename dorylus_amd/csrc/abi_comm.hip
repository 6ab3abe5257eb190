// abi_comm.hip -- C-ABI, part 3: ghost-vertex halo exchange (plan, RCCL all-to-all-v, pack / unpack for foreign
// transports), weight-gradient all-reduce + Adam, and the epoch graph (hipGraph record / replay).
#include "abi_internal.hpp"

using namespace dory;

#include <chrono>
#include <thread>

// ---------------------------------------------------------------------------------------
// In-process device transport (dory_comm_init_local): P contexts of one process on one device are each other's peers.
// One exchange = pack on the sender's comm stream -> hipMemcpyAsync device -> device into every peer's receive buffer on
// the SENDER's comm stream -> event "sent"; the receiver's comm stream waits for its peers' "sent" events, unpacks, records
// "consumed" (its receive buffer is free again) and the event the compute stream waits for.  Same stream / event structure
// as the RCCL arm, no host synchronisation with the device anywhere: what the overlapped schedule of Engine::scatterGCN +
// ghostReceiver (gcn_ops.cpp:204-282 send, :284-362 receive) looks like when copies really run beside the aggregation.
// A stream never waits for an event that is not recorded yet (hipStreamWaitEvent on an unrecorded event is a no-op, and a
// device-side wait for work a host thread has still to enqueue deadlocks with any device-wide synchronisation, e.g. a
// hipFree in a lazy allocation): every event has a progress counter its owner bumps AFTER recording, and a context reads
// the peer's counter BEFORE it makes its stream wait.  That read may block the calling host thread until the peer's host
// thread has got there (bounded: option local_timeout_ms) -- so the ranks must be driven by one host thread each, or stage
// by stage (all ranks' scatter before any rank's next gather), exactly as real ranks are.
namespace {

constexpr uint32_t LOCAL_MAX_RANKS = 16;

int local_wait_posted(dory_ctx *c, const std::atomic<uint64_t> &ctr, uint64_t want, uint32_t peer, const char *what) {
    if (ctr.load(std::memory_order_acquire) >= want) return DORY_OK;
    const int64_t lim = c->opt["local_timeout_ms"] > 0 ? c->opt["local_timeout_ms"] : 30000;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(lim);
    uint32_t spins = 0;
    while (ctr.load(std::memory_order_acquire) < want) {
        if (++spins < 2000) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((spins & 255u) == 0 && std::chrono::steady_clock::now() > deadline)
            return fail(c, DORY_ERR_COMM, "local transport: rank %u did not reach %s %llu within %lld ms (every rank needs its own host thread, or "
                        "stage-by-stage driving)", peer, what, (unsigned long long)want, (long long)lim);
    }
    return DORY_OK;
}

// an interval for dory_timing_get whose end is recorded by a later call: entry pushed now (order of `pending` = order of
// the first events), end event handed back
void timed_open(dory_ctx *c, const char *fam, hipStream_t s, hipEvent_t *end_out) {
    *end_out = nullptr;
    if (!c->timing || c->capturing) return;
    hipEvent_t a = nullptr, b = nullptr;
    if (c->ev_pool.empty()) {
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
    } else {
        a = c->ev_pool.back().first;
        b = c->ev_pool.back().second;
        c->ev_pool.pop_back();
    }
    (void)hipEventRecord(a, s);
    c->pending.push_back({fam, a, b});
    *end_out = b;
}

struct PeerPtrs { const float *p[LOCAL_MAX_RANKS]; };
__global__ __launch_bounds__(256) void local_sum_kernel(PeerPtrs pp, uint32_t P, uint64_t n, float *out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        float sum = 0.f;
        for (uint32_t q = 0; q < P; ++q) sum += pp.p[q][i];      // rank order on every rank: identical bits everywhere
        out[i] = sum;
    }
}

}  // namespace

namespace dory {

// second half of an exchange: the peers' rows have been sent (their "sent" events) -> unpack -> "consumed" + ev_b
int local_exchange_finish(dory_ctx *c) {
    dory_ctx::LocalPending &lp = c->local_pending;
    if (!lp.on) return DORY_OK;
    HaloPlan &p = c->plan[lp.dir];
    const uint64_t s = c->local_seq;
    LocalGroup &grp = *c->local;
    for (uint32_t q = 0; q < c->numNodes; ++q) {
        // only the peers that send me rows: a peer that sends me nothing does not wait for my receive buffer either, may be
        // exchanges ahead, and its event ring (two deep) would by then hold a LATER exchange's record -- one that can depend
        // on work queued behind this very wait
        if (q == c->nodeId || !p.recv_counts[q]) continue;
        dory_ctx *Q = grp.ctx[q];
        if (!Q) return fail(c, DORY_ERR_COMM, "local transport: rank %u has been destroyed", q);
        int rc = local_wait_posted(c, Q->posted_sent, s, q, "exchange");
        if (rc) return rc;
        HIPCK(c, hipStreamWaitEvent(c->comm, Q->ev_sent[s & 1], 0));
    }
    lp.on = false;     // (a peer that has not arrived leaves the second half pending: the caller may try again)
    HIPCK(c, launch_scatter_rows(lp.ghost, c->recv_buf, lp.ghost_ld, lp.w, p.d_recv_slots, p.recv_total, c->comm));
    HIPCK(c, hipEventRecord(c->ev_cons[s & 1], c->comm));
    c->posted_cons.store(s, std::memory_order_release);
    if (lp.t_kind_b) (void)hipEventRecord(lp.t_kind_b, c->comm);
    if (lp.t_halo_b) (void)hipEventRecord(lp.t_halo_b, c->comm);
    lp.t_kind_b = lp.t_halo_b = nullptr;
    HIPCK(c, hipEventRecord(c->ev_b, c->comm));
    return DORY_OK;
}

// first half: pack, push my rows into every peer's receive buffer, "sent"
static int local_exchange_send(dory_ctx *c, int dir, Tensor *src, Tensor *ghost, uint32_t w, bool deferred) {
    HaloPlan &p = c->plan[dir];
    LocalGroup &grp = *c->local;
    if (c->local_pending.on) {   // (not reached: every consumer and every exchange calls wait_halo first)
        int rc = local_exchange_finish(c);
        if (rc) return rc;
    }
    const uint64_t s = ++c->local_seq;
    dory_ctx::LocalPending &lp = c->local_pending;
    timed_open(c, "halo", c->comm, &lp.t_halo_b);
    timed_open(c, deferred ? "halo_deferred" : "halo_waited", c->comm, &lp.t_kind_b);
    HIPCK(c, launch_gather_rows(c->send_buf, src->d, src->ld, w, p.d_send_lvids, p.send_total, c->comm));
    for (uint32_t q = 0; q < c->numNodes; ++q) {
        if (q == c->nodeId || !p.send_counts[q]) continue;
        dory_ctx *Q = grp.ctx[q];
        if (!Q) return fail(c, DORY_ERR_COMM, "local transport: rank %u has been destroyed", q);
        const HaloPlan &pq = Q->plan[dir];
        if (!pq.set || pq.recv_counts.size() != c->numNodes || pq.recv_counts[c->nodeId] != p.send_counts[q])
            return fail(c, DORY_ERR_COMM, "local transport: rank %u expects %u rows from rank %u, which sends %u", q,
                        pq.set && pq.recv_counts.size() == c->numNodes ? pq.recv_counts[c->nodeId] : 0u, c->nodeId, p.send_counts[q]);
        if ((size_t)pq.recv_total * w * sizeof(float) > Q->recv_cap)
            return fail(c, DORY_ERR_COMM, "local transport: receive buffer of rank %u too small for %u-float rows", q, w);
        if (s > 1) {   // its receive buffer must have been unpacked (exchange s - 1)
            int rc = local_wait_posted(c, Q->posted_cons, s - 1, q, "the unpack of exchange");
            if (rc) return rc;
            HIPCK(c, hipStreamWaitEvent(c->comm, Q->ev_cons[(s - 1) & 1], 0));
        }
        HIPCK(c, hipMemcpyAsync(Q->recv_buf + (size_t)pq.recv_off[c->nodeId] * w, c->send_buf + (size_t)p.send_off[q] * w,
                                (size_t)p.send_counts[q] * w * sizeof(float), hipMemcpyDeviceToDevice, c->comm));
    }
    HIPCK(c, hipEventRecord(c->ev_sent[s & 1], c->comm));
    c->posted_sent.store(s, std::memory_order_release);
    lp.on = true;
    lp.dir = dir;
    lp.ghost = ghost->d;
    lp.ghost_ld = ghost->ld;
    lp.w = w;
    if (deferred) {
        c->halo_pending = true;      // wait_halo(): local_exchange_finish, then the compute stream waits for ev_b
        return DORY_OK;
    }
    int rc = local_exchange_finish(c);
    if (rc) return rc;
    HIPCK(c, hipStreamWaitEvent(c->compute, c->ev_b, 0));
    return DORY_OK;
}

// weight-gradient sum over the group: every rank adds the P gradients in rank order (identical bits on every rank)
// (`stat` = true: the three validation scalars in every context's d_stat3 instead of a weight gradient)
static int local_allreduce(dory_ctx *c, uint32_t layer, const std::string &name, float *gd, uint64_t n, bool stat = false) {
    LocalGroup &grp = *c->local;
    const uint32_t P = c->numNodes;
    if (n * sizeof(float) > c->ar_tmp_cap) return fail(c, DORY_ERR_COMM, "local transport: gradient staging buffer too small (preallocate before dory_comm_init_local)");
    const uint64_t t = ++c->local_ar_seq;
    HIPCK(c, hipEventRecord(c->ev_gready[t & 1], c->compute));
    c->posted_g.store(t, std::memory_order_release);
    PeerPtrs pp{};
    for (uint32_t q = 0; q < P; ++q) {
        dory_ctx *Q = q == c->nodeId ? c : grp.ctx[q];
        if (!Q) return fail(c, DORY_ERR_COMM, "local transport: rank %u has been destroyed", q);
        if (q != c->nodeId) {
            int rc = local_wait_posted(c, Q->posted_g, t, q, "gradient sum");
            if (rc) return rc;
            HIPCK(c, hipStreamWaitEvent(c->compute, Q->ev_gready[t & 1], 0));
        }
        if (stat) { pp.p[q] = Q->d_stat3; continue; }
        if (layer >= Q->wgrads.size()) return fail(c, DORY_ERR_COMM, "local transport: rank %u has no layer %u", q, layer);
        auto it = Q->wgrads[layer].find(name);
        if (it == Q->wgrads[layer].end() || (uint64_t)it->second.rows * it->second.ld != n)
            return fail(c, DORY_ERR_COMM, "local transport: rank %u's gradient '%s'@%u has another shape", q, name.c_str(), layer);
        pp.p[q] = it->second.d;
    }
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(local_sum_kernel, dim3(blocks), dim3(256), 0, c->compute, pp, P, n, c->ar_tmp);
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipEventRecord(c->ev_gdone[t & 1], c->compute));
    c->posted_gdone.store(t, std::memory_order_release);
    for (uint32_t q = 0; q < P; ++q) {     // nobody reads my gradient any more: the sum may replace it
        if (q == c->nodeId) continue;
        dory_ctx *Q = grp.ctx[q];
        int rc = local_wait_posted(c, Q->posted_gdone, t, q, "the end of gradient sum");
        if (rc) return rc;
        HIPCK(c, hipStreamWaitEvent(c->compute, Q->ev_gdone[t & 1], 0));
    }
    HIPCK(c, hipMemcpyAsync(gd, c->ar_tmp, n * sizeof(float), hipMemcpyDeviceToDevice, c->compute));
    return DORY_OK;
}

}  // namespace dory

extern "C" {

int dory_comm_init_local(dory_ctx *const *ctxs, uint32_t n) {
    if (!ctxs || n < 2 || n > LOCAL_MAX_RANKS) return DORY_ERR_ARG;
    for (uint32_t i = 0; i < n; ++i) {
        dory_ctx *c = ctxs[i];
        if (!c) return DORY_ERR_ARG;
        std::lock_guard<std::mutex> lock(c->mu);
        if (!c->configured || c->numNodes != n || c->nodeId != i)
            return fail(c, DORY_ERR_ARG, "comm_init_local: context %u must be configured as rank %u of %u", i, i, n);
        if (c->device != ctxs[0]->device) return fail(c, DORY_ERR_ARG, "comm_init_local: all contexts on one device");
        if (!c->plan[0].set || !c->plan[1].set) return fail(c, DORY_ERR_ARG, "comm_init_local: halo plans first (dory_partition_upload / dory_halo_plan)");
        if (!c->prealloc) return fail(c, DORY_ERR_ARG, "comm_init_local: dory_preallocate first");
    }
    auto grp = std::make_shared<LocalGroup>();
    grp->ctx.assign(ctxs, ctxs + n);
    for (uint32_t i = 0; i < n; ++i) {
        dory_ctx *c = ctxs[i];
        std::lock_guard<std::mutex> lock(c->mu);
        HIPCK(c, hipSetDevice(c->device));
        HIPCK(c, hipStreamSynchronize(c->compute));
        HIPCK(c, hipStreamSynchronize(c->comm));
        for (hipEvent_t *e : {&c->ev_sent[0], &c->ev_sent[1], &c->ev_cons[0], &c->ev_cons[1], &c->ev_gready[0], &c->ev_gready[1],
                              &c->ev_gdone[0], &c->ev_gdone[1]})
            if (!*e) HIPCK(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
        size_t need = 0;
        for (auto &m : c->wgrads)
            for (auto &kv : m) need = std::max(need, (size_t)kv.second.rows * kv.second.ld * sizeof(float));
        if (need > c->ar_tmp_cap) {
            if (c->ar_tmp) (void)hipFree(c->ar_tmp);
            c->ar_tmp = nullptr; c->ar_tmp_cap = 0;
            HIPCK(c, hipMalloc((void **)&c->ar_tmp, need));
            c->ar_tmp_cap = need;
        }
        c->posted_sent = 0; c->posted_cons = 0; c->posted_g = 0; c->posted_gdone = 0;
        c->local_seq = c->local_ar_seq = 0;
        c->local_pending = dory_ctx::LocalPending();
        c->local = grp;
    }
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
    // pack / receive buffers for the widest row any layer exchanges, now, so that no allocation (and no device-wide
    // synchronisation) happens inside an epoch
    uint32_t w = 0;
    for (uint32_t l = 0; l <= c->L; ++l) {
        w = std::max(w, pad_ld(c->dims[l]));
        if (c->gnn == DORY_GATMH && l < c->L && l < c->heads.size()) w = std::max(w, pad_ld(c->dims[l + 1] * c->heads[l]));
    }
    const size_t sb = (size_t)p.send_total * w * sizeof(float), rb = (size_t)p.recv_total * w * sizeof(float);
    if (sb > c->send_cap || rb > c->recv_cap) HIPCK(c, hipDeviceSynchronize());
    if (sb > c->send_cap) {
        if (c->send_buf) (void)hipFree(c->send_buf);
        c->send_buf = nullptr; c->send_cap = 0;
        HIPCK(c, hipMalloc((void **)&c->send_buf, sb));
        c->send_cap = sb;
    }
    if (rb > c->recv_cap) {
        if (c->recv_buf) (void)hipFree(c->recv_buf);
        c->recv_buf = nullptr; c->recv_cap = 0;
        HIPCK(c, hipMalloc((void **)&c->recv_buf, rb));
        c->recv_cap = rb;
    }
    return DORY_OK;
}

int dory_comm_set_host_transport(dory_ctx *c, dory_alltoallv_fn alltoallv, dory_allreduce_fn allreduce_sum, void *user) {
    CHECK_CTX(c);
    if ((alltoallv == nullptr) != (allreduce_sum == nullptr)) return fail(c, DORY_ERR_ARG, "set_host_transport: both callbacks or none");
    c->tx_a2a = alltoallv;
    c->tx_ar = allreduce_sum;
    c->tx_user = user;
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
    if (c->gnn == DORY_GCN && dir == DORY_BACKWARD && tf_layer(c, layer)) {
        *src = find(c, layer, "g");      // transform-first: A^T g_l needs the ghost rows of g_l
        *ghost = find(c, layer, "bgg");
    } else if (c->gnn == DORY_GCN && dir == DORY_FORWARD && layer > 0 && tf_layer(c, layer)) {
        *src = find(c, layer, "xw");     // transform-first: the already transformed (narrower) rows travel
        *ghost = find(c, layer, "fgxw");
    } else if (c->gnn == DORY_GCN) {
        if (layer == 0 || layer >= c->L) return fail(c, DORY_ERR_ARG, "halo: layer %u out of range", layer);
        if (dir == DORY_FORWARD) { *src = find(c, layer - 1, "h"); *ghost = find(c, layer, "fg"); }   // gcn_ops.cpp:205-214
        else { *src = find(c, layer, "grad"); *ghost = find(c, layer - 1, "bg"); }
    } else {
        if (layer == 0 || layer > c->L) return fail(c, DORY_ERR_ARG, "halo: layer %u out of range", layer);
        if (dir == DORY_FORWARD) {   // gat_ops.cpp:277-287
            *src = find(c, layer - 1, "z"); *ghost = find(c, layer - 1, "fg_z");
            if (layer - 1 < c->gat_nsum_valid.size()) c->gat_nsum_valid[layer - 1] = 0;   // fg_z is about to change: the kept neighbour sum no longer holds
        }
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
    if (layer == 0) c->ah0_valid = false;   // (a caller's transport writing fg@0)
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

}  // extern "C"

namespace dory {
// One all-to-all-v of rows: src rows listed in plan[dir] -> the peers' ghost tensors.  pack -> grouped
// ncclSend/ncclRecv -> unpack on the comm stream, ordered after the compute stream's work so far; the compute
// stream waits for the ghosts at once (defer == false) or when wait_halo() is next called (halo_overlap).
int exchange_rows(dory_ctx *c, int dir, Tensor *src, Tensor *ghost, bool defer) {
    HaloPlan &p = c->plan[dir];
    if (!p.set) return fail(c, DORY_ERR_ARG, "halo_exchange: no plan");
    const bool local = !c->tx_a2a && c->local;
    if (!c->tx_a2a && !local) {
        if (!c->nccl) return fail(c, DORY_ERR_COMM, "halo_exchange: dory_comm_init not called");
        if (c->nranks != (int)c->numNodes) return fail(c, DORY_ERR_COMM, "halo_exchange: communicator size != num_nodes");
    }
    if (src->ld != ghost->ld) return fail(c, DORY_ERR_ARG, "halo_exchange: row widths of source and ghost tensor differ");
    const uint32_t w = src->ld;  // padded row width travels (keeps 16-B lanes)
    const size_t sb = (size_t)p.send_total * w * sizeof(float), rb = (size_t)p.recv_total * w * sizeof(float);
    if (local && (sb > c->send_cap || rb > c->recv_cap))
        return fail(c, DORY_ERR_COMM, "local transport: exchange buffers too small for %u-float rows (peers hold their addresses: no regrowth)", w);
    if (sb > c->send_cap || rb > c->recv_cap) {   // not reached after dory_halo_plan sized them for the widest layer (a
        HIPCK(c, hipDeviceSynchronize());          // tensor uploaded with other dimensions than dory_configure's)
        if (sb > c->send_cap) {
            if (c->send_buf) (void)hipFree(c->send_buf);
            c->send_buf = nullptr; c->send_cap = 0;
            HIPCK(c, hipMalloc((void **)&c->send_buf, sb));
            c->send_cap = sb;
        }
        if (rb > c->recv_cap) {
            if (c->recv_buf) (void)hipFree(c->recv_buf);
            c->recv_buf = nullptr; c->recv_cap = 0;
            HIPCK(c, hipMalloc((void **)&c->recv_buf, rb));
            c->recv_cap = rb;
        }
    }
    // comm stream waits for the producer of `src` on the compute stream
    HIPCK(c, hipEventRecord(c->ev_a, c->compute));
    HIPCK(c, hipStreamWaitEvent(c->comm, c->ev_a, 0));
    if (local) return local_exchange_send(c, dir, src, ghost, w, defer && c->opt["halo_overlap"]);
    {
        Timed t(c, "halo", c->comm);
        Timed td(c, (defer && c->opt["halo_overlap"]) ? "halo_deferred" : "halo_waited", c->comm);   // (overlap bookkeeping: abi_internal.hpp)
        HIPCK(c, launch_gather_rows(c->send_buf, src->d, src->ld, w, p.d_send_lvids, p.send_total, c->comm));
        if (c->tx_a2a) {   // host transport: same pack / unpack / events, the bytes travel through the caller
            c->tx_send.resize((size_t)p.send_total * w);
            c->tx_recv.resize((size_t)p.recv_total * w);
            if (sb) HIPCK(c, hipMemcpyAsync(c->tx_send.data(), c->send_buf, sb, hipMemcpyDeviceToHost, c->comm));
            HIPCK(c, hipStreamSynchronize(c->comm));
            std::vector<uint64_t> sc(c->numNodes), so(c->numNodes), rc_(c->numNodes), ro(c->numNodes);
            for (uint32_t peer = 0; peer < c->numNodes; ++peer) {
                sc[peer] = (uint64_t)p.send_counts[peer] * w; so[peer] = (uint64_t)p.send_off[peer] * w;
                rc_[peer] = (uint64_t)p.recv_counts[peer] * w; ro[peer] = (uint64_t)p.recv_off[peer] * w;
            }
            if (c->tx_a2a(c->tx_user, c->tx_send.data(), sc.data(), so.data(), c->tx_recv.data(), rc_.data(), ro.data(), c->numNodes))
                return fail(c, DORY_ERR_COMM, "halo_exchange: host transport alltoallv failed");
            if (rb) HIPCK(c, hipMemcpyAsync(c->recv_buf, c->tx_recv.data(), rb, hipMemcpyHostToDevice, c->comm));
            HIPCK(c, launch_scatter_rows(ghost->d, c->recv_buf, ghost->ld, w, p.d_recv_slots, p.recv_total, c->comm));
            HIPCK(c, hipStreamSynchronize(c->comm));   // tx_recv is reused by the next exchange
        } else {
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
    }
    HIPCK(c, hipEventRecord(c->ev_b, c->comm));
    if (defer && c->opt["halo_overlap"]) c->halo_pending = true;
    else HIPCK(c, hipStreamWaitEvent(c->compute, c->ev_b, 0));
    return DORY_OK;
}
}  // namespace dory

extern "C" {

int dory_halo_exchange(dory_ctx *c, uint32_t layer, int dir) {
    CHECK_CTX(c);
    { int wrc = wait_halo(c); if (wrc) return wrc; }
    if (c->numNodes == 1) return DORY_OK;  // no ghosts
    // multi-head extension: the backward sweep exchanges dO and its per-vertex statistics itself, between its two
    // phases (dory_aggregate); the scatter stage of the reference's GAT order has nothing to ship before it
    if (c->gnn == DORY_GATMH && dir == DORY_BACKWARD) return DORY_OK;
    Tensor *src, *ghost;
    int rc = halo_tensors(c, layer, dir, &src, &ghost);
    if (rc) return rc;
    // a forward exchange of layer 0 rewrites fg@0 (a peer may have uploaded a new x): a cached ah@0 no longer holds
    if (layer == 0 && dir == DORY_FORWARD) c->ah0_valid = false;
    // consumers on the compute stream wait for the ghosts at once, or (halo_overlap) when the first of them needs
    // the ghost rows -- see wait_halo()
    return exchange_rows(c, dir, src, ghost, true);
}

// pack / unpack of any named tensor with the plan of `dir` (foreign transports, multi-context tests)
int dory_halo_pack_tensor(dory_ctx *c, uint32_t layer, const char *name, int dir, float *send_buf) {
    CHECK_CTX(c);
    { int wrc = wait_halo(c); if (wrc) return wrc; }
    Tensor *src = name ? find(c, layer, name) : nullptr;
    if (!src || (dir != 0 && dir != 1) || !c->plan[dir].set) return fail(c, DORY_ERR_ARG, "halo_pack_tensor: no tensor '%s'@%u or no plan", name ? name : "(null)", layer);
    HaloPlan &p = c->plan[dir];
    if (src->rows != c->N) return fail(c, DORY_ERR_ARG, "halo_pack_tensor: '%s' is not a per-local-vertex tensor", name);
    Timed t(c, "halo", c->compute);
    HIPCK(c, launch_gather_rows(send_buf, src->d, src->ld, src->ld, p.d_send_lvids, p.send_total, c->compute));
    return DORY_OK;
}

int dory_halo_unpack_tensor(dory_ctx *c, uint32_t layer, const char *name, int dir, const float *recv_buf) {
    CHECK_CTX(c);
    if (layer == 0) c->ah0_valid = false;   // (a caller's transport writing fg@0)
    { int wrc = wait_halo(c); if (wrc) return wrc; }
    Tensor *ghost = name ? find(c, layer, name) : nullptr;
    if (!ghost || (dir != 0 && dir != 1) || !c->plan[dir].set) return fail(c, DORY_ERR_ARG, "halo_unpack_tensor: no tensor '%s'@%u or no plan", name ? name : "(null)", layer);
    HaloPlan &p = c->plan[dir];
    if (ghost->rows != (dir == DORY_FORWARD ? c->Gsrc : c->Gdst)) return fail(c, DORY_ERR_ARG, "halo_unpack_tensor: '%s' is not a ghost tensor of that direction", name);
    Timed t(c, "halo", c->compute);
    HIPCK(c, launch_scatter_rows(ghost->d, recv_buf, ghost->ld, ghost->ld, p.d_recv_slots, p.recv_total, c->compute));
    return DORY_OK;
}

// ---------------------------------------------------------------------------------------
// The validation statistics of the last forward pass summed over all partitions: what WeightServer::updateLocalAccLoss /
// updateGlobalAccLoss do with the AccLoss records the graph servers send (src/weight-server/weightserver.cpp:190-262:
// vtcsCnt, acc and loss added up over the nodes, then "Epoch %u, acc: %.4f, loss: %.4f" on node 0).  Every rank calls it
// (a collective); every rank gets the sums.
int dory_train_stat_global(dory_ctx *c, float *acc_sum, float *loss_sum, uint32_t *val_rows) {
    CHECK_CTX(c);
    { int wrc = wait_halo(c); if (wrc) return wrc; }
    if (!c->d_stat3) return fail(c, DORY_ERR_ARG, "train_stat_global: context not created properly");
    float h[3] = {0.f, 0.f, 0.f};
    HIPCK(c, hipMemcpyAsync(h, c->d_stat, 2 * sizeof(float), hipMemcpyDeviceToHost, c->compute));
    HIPCK(c, hipStreamSynchronize(c->compute));
    h[2] = (float)c->val_rows;      // (exact below 2^24 rows per sum: Friendster's 6.6 M validation rows fit)
    if (c->numNodes > 1 && c->tx_ar) {
        if (c->tx_ar(c->tx_user, h, 3)) return fail(c, DORY_ERR_COMM, "train_stat_global: host transport allreduce failed");
    } else if (c->numNodes > 1) {
        HIPCK(c, hipMemcpyAsync(c->d_stat3, h, sizeof(h), hipMemcpyHostToDevice, c->compute));
        if (c->local) {
            int rc = local_allreduce(c, 0, "", c->d_stat3, 3, true);
            if (rc) return rc;
        } else {
            if (!c->nccl) return fail(c, DORY_ERR_COMM, "train_stat_global: dory_comm_init not called");
            NCCLCK(c, ncclAllReduce(c->d_stat3, c->d_stat3, 3, ncclFloat, ncclSum, (ncclComm_t)c->nccl, c->compute));
        }
        HIPCK(c, hipMemcpyAsync(h, c->d_stat3, sizeof(h), hipMemcpyDeviceToHost, c->compute));
        HIPCK(c, hipStreamSynchronize(c->compute));
    }
    if (acc_sum) *acc_sum = h[0];
    if (loss_sum) *loss_sum = h[1];
    if (val_rows) *val_rows = (uint32_t)(h[2] + 0.5f);
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
        if (c->numNodes > 1 && c->tx_ar) {   // host transport
            Timed t(c, "allreduce", c->compute);
            c->tx_send.resize(n);
            HIPCK(c, hipMemcpyAsync(c->tx_send.data(), g.d, n * sizeof(float), hipMemcpyDeviceToHost, c->compute));
            HIPCK(c, hipStreamSynchronize(c->compute));
            if (c->tx_ar(c->tx_user, c->tx_send.data(), n)) return fail(c, DORY_ERR_COMM, "weight_update: host transport allreduce failed");
            HIPCK(c, hipMemcpyAsync(g.d, c->tx_send.data(), n * sizeof(float), hipMemcpyHostToDevice, c->compute));
            HIPCK(c, hipStreamSynchronize(c->compute));
        } else if (c->numNodes > 1 && c->local) {
            Timed t(c, "allreduce", c->compute);
            int rc = local_allreduce(c, layer, name, g.d, n);
            if (rc) return rc;
        } else if (c->numNodes > 1) {
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

}  // extern "C"
namespace dory {
void epoch_graph_drop_locked(dory_ctx *c) {
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

}  // namespace dory
extern "C" {

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
            // on the replay stream itself: c->compute is a non-blocking stream, the legacy null stream is not ordered
            // against it (a null-stream memset could still be in flight when the replayed adam_table_kernel reads *idx)
            HIPCK(c, hipMemcpyAsync(c->d_lr_table, c->lr_table_host.data(), c->lr_table_cap * sizeof(float), hipMemcpyHostToDevice, c->compute));
            HIPCK(c, hipMemsetAsync(c->d_replay_idx, 0, sizeof(uint32_t), c->compute));
            HIPCK(c, hipStreamSynchronize(c->compute));   // lr_table_host may be resized again only after the copy
            c->lr_table_left = c->lr_table_cap;
        }
        HIPCK(c, hipGraphLaunch(c->epoch_exec, c->compute));
        c->lr_table_left -= 1;
        c->adam.epochs += 1;
    }
    return DORY_OK;
}

}  // extern "C"
