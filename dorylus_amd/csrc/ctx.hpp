// ctx.hpp -- internal state behind the C-ABI (include/dorylus_hip.h).
// MI355X / gfx950 only; no CUDA shims, no dual paths.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dorylus_hip.h"

namespace dory {

// Device tensor: row-major fp32, leading dimension padded to 32 floats (one
// 128-B line) so that every row starts on a cache-line boundary and float4
// lanes never straddle rows (602 -> 608).  Host layout stays dense.
struct Tensor {
    uint64_t rows = 0;
    uint32_t cols = 0;
    uint32_t ld = 0;
    float *d = nullptr;
    bool owned = true;  // false: alias of another allocation (GAT "A" = csc values)
    size_t bytes() const { return (size_t)rows * ld * sizeof(float); }
};

inline uint32_t pad_ld(uint32_t cols) { return cols <= 1 ? cols : (cols + 31u) & ~31u; }

struct Timing {
    double total_ms = 0;
    uint64_t launches = 0;
};

struct HaloPlan {
    bool set = false;
    std::vector<uint32_t> send_counts, recv_counts;  // per peer (rows)
    std::vector<uint32_t> send_off, recv_off;        // prefix sums
    uint32_t send_total = 0, recv_total = 0;
    uint32_t *d_send_lvids = nullptr;  // concat over peers
    uint32_t *d_recv_slots = nullptr;  // concat over peers
};

// K1b: source-blocked adjacency (per-block CSR over the virtual source space [local;ghost])
struct BlockedAdj {
    uint32_t nb = 0;        // number of source blocks (multiple of 8: one per XCD per round)
    uint32_t SB = 0;        // rows per block
    uint32_t row_bytes = 0; // slab bytes per row the block size was chosen for
    uint32_t npos = 0;      // destination positions = leading dimension of boff minus 1 (N, or more with `perm`)
    uint32_t nb_local = 0;  // blocks [0, nb_local) contain local source rows only
    uint32_t nghost = 0;    // K1s layout: ghost source rows (blocks [nb_local, nb) contain only those)
    uint32_t rows_per_group = 0;   // K1s layout: the R its positions were dealt for
    // K1s layout (build_blocked_sweep): destination positions are a degree-balanced deal of the rows (rows of very high
    // degree cut into pieces), source rows are spread over the blocks by a random permutation (local / ghost rows apart)
    uint32_t *perm = nullptr;        // npos: position -> row, 0xFFFFFFFF = empty; nullptr = identity
    uint32_t *otgt = nullptr;        // npos: where a position's sum goes: the row, or 0x80000000 | slot for a piece of a split row
    uint32_t *split_rows = nullptr;  // 3 words per split row: row, first slot, pieces
    uint32_t nsplit = 0, nslots = 0;
    uint64_t *bbase = nullptr;  // nb+1: first edge of each block
    uint32_t *boff = nullptr;   // [nb][N+1]: row offsets inside the block
    uint32_t *bidx = nullptr;   // nnz: source row (virtual id), block-major / row-minor / edge order
    float *bval = nullptr;      // nnz
    uint2 *bent = nullptr;      // K1s layout: (bidx, bval bits) interleaved instead of the two arrays (nnz + 2 entries, 16-byte aligned pairs)
    // (block,row) segments longer than BLK_SEG_CLAMP edges (hubs): K1b stops there, the remainder is cut into chunks
    // of BLK_SEG_CHUNK edges for workgroup-per-chunk kernels (spmm.hip); all empty on graphs without such segments
    uint32_t seg_clamp = 0;             // 0: no long segments
    uint32_t nsegs = 0, nchunks = 0;
    uint32_t *seg_row = nullptr;        // nsegs: destination row
    uint32_t *seg_blk = nullptr;        // nsegs: source block
    uint32_t *seg_chunk_ptr = nullptr;  // nsegs+1
    uint32_t *seg_chunks = nullptr;     // 6 words per chunk: row, block, e0 (lo, hi), e1 (lo, hi): absolute positions in bidx
};
constexpr uint32_t BLK_SEG_CLAMP = 2048, BLK_SEG_CHUNK = 4096;
// K1's long rows (hubs): the part of a row beyond LONG_ROW_CLAMP edges, in chunks of LONG_ROW_CHUNK edges
constexpr uint32_t LONG_ROW_CLAMP = 8192, LONG_ROW_CHUNK = 4096;
struct LongRowsHost {                    // plan_long_rows output
    std::vector<uint32_t> rows;          // long rows
    std::vector<uint32_t> row_chunk_ptr; // rows+1: first chunk of each
    std::vector<uint32_t> chunks;        // 6 words per chunk: row, 0, e0 (lo, hi), e1 (lo, hi)  (= struct LongChunk)
};
// K1 beside an exchange in flight (round 6): every row's edges regrouped local sources first, ghost sources after (stable), so
// that ONE pass over all rows sums the local part while the ghost rows are still travelling and a second pass adds the rest --
// the row gather's counterpart of K1s's local-source blocks.  The interior / boundary ROW split it replaces hides nothing on
// a graph where nearly every row has a ghost neighbour (mean degree 25, 15 % remote edges: 98 % of the rows).
struct EdgeSplit {
    uint32_t *idx = nullptr;   // nnz: source ids, local ones first within every row
    float *val = nullptr;      // nnz
    uint64_t *mid = nullptr;   // N: first ghost-source edge of the row
};
struct LongRowsDev {
    uint32_t nrows = 0, nchunks = 0;
    uint32_t *rows = nullptr, *row_chunk_ptr = nullptr, *chunks = nullptr;
};
// In-process device transport (dory_comm_init_local, abi_comm.hip): the contexts of one process that are each other's
// peers.  Peers read each other's plans, buffers, events and progress counters; nothing else.
struct LocalGroup {
    std::mutex mu;                    // membership only (a context leaving at dory_destroy)
    std::vector<dory_ctx *> ctx;      // by rank; nullptr once destroyed
};
struct AdamState {
    float lr = 0.01f;
    unsigned epochs = 1;  // AdamOptimizer ctor calls nextIteration() once
    float lr_t = 0.f;
};

}  // namespace dory

struct dory_ctx {
    int device = 0;
    std::mutex mu;
    std::string err;
    hipStream_t compute = nullptr, comm = nullptr;
    bool own_compute = false, own_comm = false;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;  // cross-stream ordering
    bool halo_pending = false;  // ghosts of the last exchange are still landing (comm stream, ev_b)

    // model
    bool configured = false;
    int gnn = DORY_GCN;
    uint32_t L = 0;
    std::vector<uint32_t> dims;
    uint32_t globalV = 0, nodeId = 0, numNodes = 1;
    std::vector<uint32_t> heads;   // multi-head GAT extension: heads per layer

    // graph (Graph, graph/graph.hpp:60-99)
    bool has_graph = false;
    uint32_t N = 0, Gsrc = 0, Gdst = 0;
    uint64_t nnz_in = 0, nnz_out = 0;
    uint64_t *colPtr = nullptr, *rowPtr = nullptr;
    uint32_t *rowIdx = nullptr, *colIdx = nullptr;
    float *cscVal = nullptr, *csrVal = nullptr, *norm = nullptr;
    // longest-row-first schedules for the SpMM (built at upload)
    uint32_t *orderIn = nullptr, *orderOut = nullptr;
    bool skewIn = false, skewOut = false;   // max degree > 8 x mean: K1 walks the rows longest first (option spmm_order = 1)
    // K1 under a halo exchange in flight: rows whose sources are all local ("interior", first nInt entries) run
    // first, the rows that read ghost rows after the exchange; both parts longest row first
    uint32_t *splitIn = nullptr, *splitOut = nullptr;   // N entries: interior rows, then boundary rows
    uint32_t nIntIn = 0, nIntOut = 0;
    dory::LongRowsDev longIn, longOut;          // K1: hub rows of forwardAdj / backwardAdj
    bool agg_static_ghosts = false;             // the aggregation being issued reads ghost rows no exchange writes (layer 0 forward)
    dory::EdgeSplit esIn, esOut;                // K1: local-first copies of forwardAdj / backwardAdj (GCN partitions with ghosts)
    // K1b blocked copies of forwardAdj / backwardAdj (built on first use) + partial buffer
    dory::BlockedAdj blkIn, blkOut;
    // multi-head GAT contexts with layers on both sides of 128 floats: a second pair, blocked for the 256-B slabs of the narrow
    // layers (K1b's windows hold a fixed number of BYTES: 24 blocks of 512-B rows, 16 of 256-B rows at Reddit size)
    dory::BlockedAdj blkIn16, blkOut16;
    bool blkIn16_built = false, blkOut16_built = false;
    dory::BlockedAdj swpIn, swpOut;             // K1s layouts (built on first use when spmm_variant == 2)
    bool swpIn_built = false, swpOut_built = false, swpIn_na = false, swpOut_na = false;
    uint32_t swpIn_want_nb = 0, swpOut_want_nb = 0;   // the spmm_blk_nb the layouts were built for
    bool blkIn_built = false, blkOut_built = false;
    bool blkIn_na = false, blkOut_na = false;   // K1b not applicable (too many source blocks): use K1
    uint32_t cus_per_xcd = 32;                  // K1s: workgroups per sweep
    // K1s's placement assumption, checked once per context (dory_create: HW_REG_XCC_ID of 2048 probe workgroups -- equal
    // id & 7 => same XCD, the eight residues on eight XCDs).  When it does not hold (a CPX / NPS-partitioned or CU-masked device) the gates
    // synchronise workgroups that do not share an L2: the first idempotent K1s launch is then run gated and ungated, timed,
    // and the faster form kept (xcd_policy: -1 not decided, 0 gated, 8 ungated = the SWEEP flag)
    bool xcd_mapping_ok = true;
    uint32_t xcd_count = 8;
    int xcd_policy = -1;
    float xcd_gated_ms = 0.f, xcd_ungated_ms = 0.f;
    uint32_t *sweep_stat = nullptr;             // K1s: SWEEP_STAT_WORDS device words that outlive the launches: gate timeouts, back-off horizon of
                                                // the launches that run alone, ungated launches, back-off horizon of the launches
                                                // beside an exchange, launch number (bumped on the device: hipGraph replays advance
                                                // it), workgroups of the launch in flight that have left
    // GAT: per-destination edge factors written by dory_apply_edge are valid for these layers
    std::vector<char> gat_arow_valid, gat_drow_valid;
    // GAT prototype, "gat_lazy_edge_tensors": the per-edge tensors az / A / dA hold one value per DESTINATION; the stages keep the
    // per-vertex values (azrow / arow / drow) and write the per-edge copies only when somebody asks for them
    std::vector<char> gat_azrow_valid;              // azrow@l describes az@l (false after a caller uploaded az)
    std::vector<char> gat_az_stale, gat_dA_stale;   // per layer: az@l / dA@l are behind azrow@l / drow@l
    int gat_A_stale_layer = -1;                     // "A" (= forwardAdj.values) is behind arow@this layer
    std::vector<char> gat_nsum_valid;    // GAT prototype: "nsum"@l holds the unweighted neighbour sum of the current z / fg_z (forward -> backward)
    bool gat_ones_set = false;           // "ones"@0 filled
    bool last_spmm_unit = false;         // the last spmm() gathered with unit weights and a per-row factor (K1s / K1b), not with per-edge values (K1)
    std::vector<char> gatmh_fwd_swept;   // multi-head GAT: the forward of this layer ran on the sweep skeleton ("op", "dpos" are current)
    float *partial = nullptr;
    size_t partial_bytes = 0;

    // tensors / weights
    bool prealloc = false;
    bool ah0_valid = false;      // option gcn_cache_ah0: ah@0 holds the aggregate of the current x / fg@0 / adjacency
    uint64_t ah0_skips = 0;      // layer-0 aggregations answered from it
    std::vector<std::map<std::string, dory::Tensor>> tensors;   // [layer][name]
    std::vector<std::map<std::string, dory::Tensor>> weights;   // "w", "a_i"
    std::vector<std::map<std::string, dory::Tensor>> wgrads;    // same names
    std::vector<std::map<std::string, dory::Tensor>> adam_m, adam_v;
    dory::AdamState adam;

    // scratch
    float *scratch = nullptr;
    size_t scratch_bytes = 0;
    float *d_stat = nullptr;  // [acc_sum, loss_sum]
    float *d_stat3 = nullptr; // dory_train_stat_global: [acc_sum, loss_sum, validation rows] of this partition, then of all
    uint32_t val_rows = 0;

    // halo
    dory::HaloPlan plan[2];
    float *send_buf = nullptr, *recv_buf = nullptr;
    size_t send_cap = 0, recv_cap = 0;
    void *nccl = nullptr;  // ncclComm_t
    // host transport (dory_comm_set_host_transport): callbacks + host staging
    dory_alltoallv_fn tx_a2a = nullptr;
    dory_allreduce_fn tx_ar = nullptr;
    void *tx_user = nullptr;
    std::vector<float> tx_send, tx_recv;
    int rank = 0, nranks = 1;
    // in-process device transport (dory_comm_init_local): rows travel device -> device on the sender's comm stream, ordered
    // by cross-context events; a context only ever waits (on the device) for events its peer has ALREADY recorded -- the
    // host side checks the peer's progress counter first -- so no stream depends on a host call still to come
    std::shared_ptr<dory::LocalGroup> local;
    hipEvent_t ev_sent[2] = {nullptr, nullptr};    // [seq & 1]: my rows of exchange seq have landed in the peers' receive buffers
    hipEvent_t ev_cons[2] = {nullptr, nullptr};    // [seq & 1]: my receive buffer of exchange seq has been unpacked
    hipEvent_t ev_gready[2] = {nullptr, nullptr};  // [seq & 1]: my weight gradient of sum seq is final
    hipEvent_t ev_gdone[2] = {nullptr, nullptr};   // [seq & 1]: I have read every peer's gradient of sum seq
    std::atomic<uint64_t> posted_sent{0}, posted_cons{0}, posted_g{0}, posted_gdone{0};   // last seq whose event is recorded
    uint64_t local_seq = 0, local_ar_seq = 0;      // exchanges / gradient sums issued so far
    struct LocalPending {                           // second half of a deferred exchange (wait for the peers' rows, unpack)
        bool on = false;
        int dir = 0;
        float *ghost = nullptr;
        uint32_t ghost_ld = 0, w = 0;
        hipEvent_t t_halo_b = nullptr, t_kind_b = nullptr;   // timing: end events of the "halo" / "halo_deferred|waited" intervals
    } local_pending;
    float *ar_tmp = nullptr;          // gradient sum before it replaces the local gradient
    size_t ar_tmp_cap = 0;

    // epoch graph (hipGraph replay of one captured epoch; single partition)
    bool capturing = false;
    hipGraph_t epoch_graph = nullptr;
    hipGraphExec_t epoch_exec = nullptr;
    float *d_lr_table = nullptr;       // Adam step size of each replay of the current launch batch
    uint32_t lr_table_cap = 0;
    uint32_t lr_table_left = 0;        // prepared entries not yet consumed by a replay
    uint32_t *d_replay_idx = nullptr;  // bumped by the last node of the graph
    std::vector<float> lr_table_host;

    // options / timing
    std::map<std::string, int64_t> opt;
    bool timing = false;
    std::map<std::string, dory::Timing> times;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    struct Pending { std::string fam; hipEvent_t a, b; };
    std::vector<Pending> pending;
};

namespace dory {

// ---- kernel launchers (one translation unit per family) ---------------------
// K1  CSR/CSC SpMM with fused self term:
//   out[v,:] = self[v]*xl[v,:] + sum_e val[e] * row(idx[e])   (self = norm | 1 | none)
struct SpmmArgs {
    uint32_t N;             // rows (local vertices)
    uint32_t F;             // logical feature width
    uint32_t ld;            // leading dimension of xl, xg, out (all equal)
    const uint64_t *ptr;    // N+1
    const uint32_t *idx;    // nnz, local ids; >= N means ghost row idx-N
    const float *val;       // nnz
    const float *self_scale;// N or nullptr
    int self_mode;          // 0: no self term, 1: self_scale[v]*xl[v], 2: 1*xl[v]
    const float *xl;        // N x ld
    const float *xg;        // G x ld (may be nullptr when no ghosts)
    float *out;             // N x ld
    int accumulate;         // 1: out += result (GAT backward second pass)
    uint32_t row_clamp;     // K1: edges of a row beyond this many are left to the long-row kernels (0 = no limit)
    uint32_t rows;          // K1: rows of this launch = the first `rows` entries of `order` (0 = all N rows)
    const uint32_t *order;  // optional row schedule (longest first) or nullptr
    const uint64_t *ptr_end;// K1: end of every row's edge range (nullptr: ptr[v + 1]); with accumulate == 2 the sum STARTS from out[v]
};
hipError_t launch_spmm(const SpmmArgs &a, int variant, int slab, hipStream_t s);

void plan_long_rows(const uint64_t *ptr, uint32_t N, LongRowsHost *out);
hipError_t launch_spmm_long_rows(const SpmmArgs &a, const LongRowsDev &L, float *partial /*nchunks x ld*/, hipStream_t s);

hipError_t build_blocked(const uint64_t *ptr, const uint32_t *idx, const float *val, uint32_t N, uint32_t NG,
                         uint64_t nnz, uint32_t want_nb /*0 = auto*/, uint32_t row_bytes, BlockedAdj *out,
                         hipStream_t s, uint64_t window_bytes = 0 /*0 = K1b's windows*/);
uint32_t plan_blocks(uint32_t NG, uint32_t want_nb, uint32_t row_bytes, uint64_t window_bytes = 0);
hipError_t launch_spmm_blocked_part(const SpmmArgs &a, const BlockedAdj &B, float *partial, int group, bool unit,
                                    uint32_t b_lo, uint32_t b_hi, hipStream_t s);
// the long (block,row) segments' remainder added into their partial rows (between the part launches and the reduce)
hipError_t launch_spmm_blocked_long_segments(const SpmmArgs &a, const BlockedAdj &B, float *partial, bool unit,
                                             float *chunk_partial /*nchunks x ld*/, hipStream_t s);
hipError_t launch_spmm_blocked_reduce(const SpmmArgs &a, const BlockedAdj &B, const float *partial,
                                      const float *row_scale, hipStream_t s);
void free_blocked(BlockedAdj *B);
// K1s: the register-accumulating sweep over its own even layout of the blocked adjacency (spmm.hip)
hipError_t build_blocked_sweep(const uint64_t *ptr, const uint32_t *idx, const float *val, uint32_t N, uint32_t NG,
                               uint64_t nnz, uint32_t want_nb, uint32_t row_bytes, uint64_t window_bytes, int R,
                               BlockedAdj *out, hipStream_t s, uint32_t layout = 3 /* 1: spread sources, 2: deal rows by degree */,
                               uint32_t sweep_tiles = 32 /* workgroups per sweep and XCD the deal is made for */,
                               uint32_t loader_relief = 0 /* rows per sweep kept off the two lane groups of every workgroup's wave 0 */);
int sweep_pick_r(uint32_t N, int group, uint32_t G, int force_r = 0 /* option spmm_sweep_rows: 0 = by fill */, int max_r = 10);
// host/sweep_deal.cpp: rows per lane group of the K1s layout and the position of every (sorted) item
bool sweep_deal_plan(uint32_t nl, uint32_t R, uint32_t sweep_tiles, std::vector<uint32_t> *cap, uint32_t *npos, uint32_t loader_relief = 0);
bool sweep_deal_positions(uint32_t nl, uint32_t R, const std::vector<uint32_t> &cap, uint32_t *pos, uint32_t loader_lo = 0);
bool sweep_deal_balanced(uint32_t nl, uint32_t R, const std::vector<uint32_t> &cap, const uint64_t *w, uint32_t *pos);
bool sweep_supported(const SpmmArgs &a, const BlockedAdj &B, int group);
// per-context knobs and state of the K1s launches (nothing process-wide)
struct SweepCtl {
    int force_r = 0;            // option spmm_sweep_rows
    int pair = -1;              // option spmm_sweep_pair
    bool loader = true;         // option spmm_sweep_loader: wave 0 of a workgroup copies the next step's entries for all (32-lane launches)
    uint32_t *stat = nullptr;   // SWEEP_STAT_WORDS device words that outlive the launches (dory_ctx::sweep_stat)
};
constexpr int SWEEP_STAT_WORDS = 8;
size_t sweep_scratch_bytes(const BlockedAdj &B, uint32_t ld, int group, uint32_t G, uint32_t nblocks, int force_r = 0);
hipError_t launch_spmm_sweep(const SpmmArgs &a, const BlockedAdj &B, int group, const float *row_scale, uint32_t cus_per_xcd,
                             uint32_t b_lo, uint32_t b_hi, uint32_t *done /* sweep_scratch_bytes() */, hipStream_t s,
                             const SweepCtl &ctl, uint32_t flags = 0, float *split_partial = nullptr /* B.nslots x ld floats */,
                             uint32_t reserve_cus = 0 /* CUs per XCD left to concurrent kernels */);
hipError_t launch_spmm_sweep_combine(const SpmmArgs &a, const BlockedAdj &B, const float *row_scale, const float *split_partial,
                                     hipStream_t s);
hipError_t launch_occupy_cus(uint32_t workgroups, uint64_t usec, hipStream_t s);   // diagnostic (dory_debug_occupy_cus)
hipError_t launch_xcd_probe(uint32_t *xcc /*grid words*/, uint32_t grid, hipStream_t s);   // HW_REG_XCC_ID of every probe workgroup
size_t blocked_partial_bytes(const SpmmArgs &a, const BlockedAdj &B);
hipError_t launch_spmm_blocked(const SpmmArgs &a, const BlockedAdj &B, float *partial, int group /*8|16|32 lanes per row*/,
                               const float *row_scale /*nullable: unit edge weights, per-row factor*/, hipStream_t s);

// K2  fp32 MFMA GEMM  C = op(A) op(B) with fused epilogues
enum GemmEpilogue { EPI_NONE = 0, EPI_TANH = 1 /* also writes tanh(C) to C2 */ };
struct GemmArgs {
    int ta, tb;             // op(A): M x K, op(B): K x N
    uint32_t M, N, K;
    const float *A; uint32_t lda;
    const float *B; uint32_t ldb;
    float *C; uint32_t ldc;
    float *C2; uint32_t ldc2;   // EPI_TANH: h = tanh(z)
    int epilogue;
    // prologue on A (NN / TN only): A'[i,k] = A[i,k] * (1 - tanh(Zp[i,k])^2)
    const float *Zp; uint32_t ldz;
};
hipError_t launch_gemm(const GemmArgs &g, float *scratch, size_t scratch_bytes, hipStream_t s);
size_t gemm_scratch_bytes(uint32_t M, uint32_t N, uint32_t K);

// tanh of the transform's epilogue (K2), of the stand-alone activation and of K3's backward -- one function for all three, so
// that h and the 1 - tanh^2 of the backward pass are the same number.  libm's tanhf is ~40 vector instructions (40 M of the
// 53.6 M of a 602->128 launch, round 4); this one is 13: an odd polynomial below 0.35 (Taylor through x^11: the next term is
// 1.2e-8 relative there) and 1 - 2 / (1 + exp(2x)) above (v_exp_f32 + v_rcp_f32; at most ~3e-7 relative at the crossover,
// where the subtraction costs most).  tests/test_gpu_parity.py::test_tanh_matches_libm: <= 1e-6 relative against the oracle.
__device__ __forceinline__ float dory_tanh(float x) {
#ifdef DORY_TANH_LIBM
    return tanhf(x);
#endif
    const float ax = fabsf(x), x2 = x * x;
    float p = fmaf(x2, 0.0088632355f, -0.021869488f);          // 1382/155925, -62/2835
    p = fmaf(x2, p, 0.053968254f);                              // 17/315
    p = fmaf(x2, p, -0.13333333f);                              // -2/15 ... signs alternate: tanh x = x - x^3/3 + 2x^5/15 - 17x^7/315 + 62x^9/2835 - 1382x^11/155925
    p = fmaf(x2, p, 0.33333334f);
    const float small = fmaf(-x * x2, p, x);                    // x - x^3 (1/3 - x^2 (2/15 - ...))
    const float t = __builtin_amdgcn_exp2f(ax * 2.885390081777927f);   // exp(2|x|); inf for |x| > 44: 2/(1+inf) = 0
    const float big = copysignf(1.f - 2.f * __builtin_amdgcn_rcpf(1.f + t), x);
    return ax < 0.35f ? small : big;
}

// K3/K4 elementwise + loss
hipError_t launch_tanh_backward(uint64_t rows, uint32_t cols, const float *aTg, uint32_t lda,
                                const float *z, uint32_t ldz, float *g, uint32_t ldg, hipStream_t s);
hipError_t launch_tanh_forward(uint64_t rows, uint32_t cols, const float *z, uint32_t ldz, float *h, uint32_t ldh,
                               hipStream_t s);
// softmax + validation stats + maskout quirk + (p - lab)/denom  (CPU_comm.cpp:108-122)
size_t softmax_xent_scratch_bytes(uint32_t cols, uint32_t val_rows);
hipError_t launch_softmax_xent(uint32_t rows, uint32_t cols, const float *z, uint32_t ldz,
                               const float *lab, uint32_t ldl, float *d, uint32_t ldd,
                               float denom, uint32_t val_stt, uint32_t val_end,
                               uint64_t mask_first, uint64_t mask_count, float *stat,
                               float *stat_partial /* 2 * ceil(val_rows/256) floats */, hipStream_t s);
// predictGAT: grad = softmax(z) - lab
hipError_t launch_softmax_sub(uint32_t rows, uint32_t cols, const float *z, uint32_t ldz,
                              const float *lab, uint32_t ldl, float *out, uint32_t ldo, hipStream_t s);
hipError_t launch_fill_uniform(float *d, uint64_t rows, uint32_t cols, uint32_t ld, uint64_t seed,
                               float lo, float hi, hipStream_t s);
hipError_t launch_fill_uniform_ids(float *d, uint64_t rows, uint32_t cols, uint32_t ld,
                                   const uint32_t *row_ids, uint64_t seed, float lo, float hi,
                                   hipStream_t s);
hipError_t launch_onehot(float *d, uint64_t rows, uint32_t cols, uint32_t ld, const uint32_t *labels,
                         hipStream_t s);
hipError_t launch_pad_copy(float *dst, uint32_t ldd, const float *src, uint32_t lds, uint64_t rows,
                           uint32_t cols, hipStream_t s);
hipError_t launch_row_axpy(float *out, const float *S, const float *rs, const float *x /* nullptr: out += rs * S */, uint64_t rows, uint32_t ld,
                           hipStream_t s);

// K5 GAT edge kernels
hipError_t launch_edge_forward_gat(uint32_t N, uint32_t F, const uint64_t *colptr, const float *z,
                                   uint32_t ldz, const float *a, float *az, float *A,
                                   float *arow /*N: the column's A value*/, hipStream_t s,
                                   float *azrow = nullptr /*N: the column's az value; az == A == nullptr: per-edge copies left to launch_expand_rows_to_edges*/);
hipError_t launch_expand_rows_to_edges(uint32_t N, const uint64_t *colptr, const float *row, float *out, hipStream_t s);
hipError_t launch_edge_backward_gat(uint32_t N, uint32_t F, const uint64_t *colptr, const float *grad,
                                    uint32_t ldg, const float *az, const float *a, float *dA,
                                    float *cw /*N: deg(v)*dLRelu_v*/, float *drow /*N: the column's dA value*/,
                                    hipStream_t s, const float *azrow = nullptr /*N: read instead of az; dA may then be nullptr*/);
hipError_t launch_rowdot(uint32_t N, uint32_t F, const float *X, uint32_t ld, const float *r, float *y,
                         hipStream_t s);
hipError_t launch_colsum_w(uint32_t N, uint32_t F, const float *X, uint32_t ld, const float *w,
                           float *partial, size_t partial_bytes, float *out, hipStream_t s);

// multi-head GAT extension (csrc/gat_mh.hip)
hipError_t launch_gatmh_scores(uint32_t N, uint32_t K, uint32_t D, const float *z, uint32_t ldz, const float *a_l,
                               const float *a_r, float *el, float *er, uint32_t ldk, hipStream_t s);
hipError_t launch_gatmh_forward(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const uint64_t *colptr,
                                const uint32_t *rowidx, const float *z, const float *el, const float *er, float *o,
                                float *m, float *den, hipStream_t s);
// source-blocked forward (statistics pass, weighted sum over K1b's blocked adjacency, reduce + self edge)
hipError_t launch_gatmh_forward_blocked(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk,
                                        const uint64_t *colptr, const uint32_t *rowidx, const BlockedAdj &B,
                                        const float *z, const float *zg /*ghost rows (ids >= N) or nullptr*/,
                                        const float *el, const float *elg, const float *er, float *o, float *m,
                                        float *den, float *partial /*nb x N x ld*/, bool ghosts, hipStream_t s,
                                        float *stat_partial = nullptr /*2 x nb x N x ldk floats: online softmax per block, no statistics pass*/,
                                        const float *a_l = nullptr /*K x D: source scores formed from the gathered rows*/);
// source-blocked backward in two phases (a partitioned run exchanges the ghost rows of dO and st4 in between):
// dst = t, der, st4 = (er, m, 1/den, t) per local (v,k); src = del, dz.  lds4: float4 stride of the st4 rows.
bool gatmh_backward_blocked_ok(uint32_t K, uint32_t D, uint32_t ld);
hipError_t launch_gatmh_backward_blocked_dst(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk,
                                             const BlockedAdj &Bin, const float *z, const float *zg, const float *el,
                                             const float *elg, const float *er, const float *m, const float *den,
                                             const float *d_o, float *t, float *der, float *partial, float4 *st4,
                                             uint32_t lds4, bool ghosts, hipStream_t s,
                                             const float *a_l = nullptr /*K x D: source scores formed from the gathered rows*/);
hipError_t launch_gatmh_backward_blocked_src(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk,
                                             const BlockedAdj &Bout, const float *z, const float *el, const float *d_o,
                                             const float *dog, const float4 *st4, const float4 *stg, uint32_t lds4,
                                             const float *der, const float *a_l, const float *a_r, float *del, float *dz,
                                             float *partial /*nb x N x (ld + K) floats*/, bool ghosts, hipStream_t s);
// the edge passes on K1s's skeleton (csrc/gat_mh_sweep.hip): register-resident sums over all source blocks of the sweep layout,
// single-pass softmax against a per-(v,k) upper-bound shift; the destination side of the backward pass needs no edges
int gatmh_sweep_hl(uint32_t K, uint32_t D, uint32_t ld);   // lanes per head; 0 = shape not covered (blocked kernels)
int gatmh_sweep_rows(const BlockedAdj &S, int group, int HL, int pass /*0 forward, 1 source side*/);   // rows per lane group of a launch
size_t gatmh_sweep_scratch_bytes(const BlockedAdj &S, uint32_t N, uint32_t ld, uint32_t ldk);
hipError_t launch_gatmh_sweep_begin(uint32_t N, uint32_t G, uint32_t K, uint32_t ld, uint32_t ldk, const BlockedAdj &S, const float *el,
                                    const float *elg, float *scratch, hipStream_t s);
hipError_t launch_gatmh_forward_sweep_part(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const BlockedAdj &S,
                                           const float *z, const float *zg, const float *er, const float *a_l, float *o, float *op,
                                           float *scratch, uint32_t cus, uint32_t b_lo, uint32_t b_hi, bool accumulate, uint32_t *done,
                                           const SweepCtl &ctl, uint32_t flags, hipStream_t s, const float *el, const float *elg /* the sources' scores (local, ghost rows) */);
hipError_t launch_gatmh_forward_sweep_finish(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const uint64_t *colptr,
                                             const uint32_t *rowidx, const BlockedAdj &S, const float *z, const float *zg, const float *el,
                                             const float *elg, const float *er, float *o, float *op, float *m, float *den, float *dpos,
                                             float *scratch, hipStream_t s);
hipError_t launch_gatmh_dst_rowwise(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const float *d_o, const float *o,
                                    const float *op, const float *dpos, const float *er, const float *m, const float *den, float *t,
                                    float *der, float4 *st4, uint32_t lds4, hipStream_t s);
size_t gatmh_src_sweep_scratch_bytes(const BlockedAdj &S, uint32_t N, uint32_t G, uint32_t K, uint32_t ld, uint32_t ldk);
hipError_t launch_gatmh_src_sweep_begin(uint32_t N, uint32_t G, uint32_t K, uint32_t ld, uint32_t ldk, const BlockedAdj &S,
                                        const float4 *st4, const float4 *stg, uint32_t lds4, float *scratch, hipStream_t s);
hipError_t launch_gatmh_src_sweep_part(uint32_t N, uint32_t G, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const BlockedAdj &S,
                                       const float *d_o, const float *dog, const float *el, float *dz, float *scratch, uint32_t cus,
                                       uint32_t b_lo, uint32_t b_hi, bool accumulate, uint32_t *done, const SweepCtl &ctl, uint32_t flags,
                                       hipStream_t s);
hipError_t launch_gatmh_src_sweep_finish(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const BlockedAdj &S, const float *z,
                                         const float *el, const float *d_o, const float *der, const float *a_l, const float *a_r, float *del,
                                         float *dz, float *scratch, hipStream_t s);
// row-wise backward (single partition; feature-per-lane or (edge, head, piece)-per-lane kernels by shape)
hipError_t launch_gatmh_backward(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const uint64_t *colptr,
                                 const uint32_t *rowidx, const uint64_t *rowptr, const uint32_t *colidx,
                                 const float *z, const float *el, const float *er, const float *m, const float *den,
                                 const float *d_o, const float *a_l, const float *a_r, float *t, float *del,
                                 float *der, float *dz, float *da_l, float *da_r, float *scratch,
                                 size_t scratch_bytes, hipStream_t s);
hipError_t launch_gatmh_dattn(uint32_t N, uint32_t K, uint32_t D, uint32_t ld, uint32_t ldk, const float *z,
                              const float *del, const float *der, float *da_l, float *da_r, float *scratch,
                              size_t scratch_bytes, hipStream_t s);
hipError_t launch_gatmh_elu(uint64_t rows, uint32_t cols, const float *o, uint32_t ldo, float *h, uint32_t ldh, hipStream_t s);
hipError_t launch_gatmh_elu_bwd(uint64_t rows, uint32_t cols, const float *dh, uint32_t lddh, const float *o,
                                uint32_t ldo, float *d_o, uint32_t lddo, hipStream_t s);
hipError_t launch_gatmh_head_mean(uint64_t rows, uint32_t K, uint32_t C, const float *o, uint32_t ldo, float *logits,
                                  uint32_t ldl, hipStream_t s);
hipError_t launch_gatmh_head_expand(uint64_t rows, uint32_t K, uint32_t C, const float *dl, uint32_t lddl, float *d_o,
                                    uint32_t lddo, hipStream_t s);

// K6 halo pack / unpack
hipError_t launch_gather_rows(float *dst, const float *src, uint32_t ld, uint32_t cols,
                              const uint32_t *rows, uint32_t n, hipStream_t s);
hipError_t launch_scatter_rows(float *dst, const float *src, uint32_t ld, uint32_t cols,
                               const uint32_t *rows, uint32_t n, hipStream_t s);

// K7 Adam
hipError_t launch_adam(float *w, const float *g, float *m, float *v, uint64_t n, float lr_t,
                       hipStream_t s);
hipError_t launch_adam_table(float *w, const float *g, float *m, float *v, uint64_t n, const float *lr_table,
                             const uint32_t *idx, hipStream_t s);
hipError_t launch_bump_counter(uint32_t *idx, hipStream_t s);

}  // namespace dory
