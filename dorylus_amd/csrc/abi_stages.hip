// abi_stages.hip -- C-ABI, part 2: the stage dispatch that replaces the reference's Engine::aggregate* bodies and
// ResourceComm::NNCompute (aggregate, apply_vertex, apply_edge, predict, validation statistics).
#include "abi_internal.hpp"

namespace dory {
// ---------------------------------------------------------------------------------------
// K1b bookkeeping: (re)build the source-blocked copy of one adjacency for `group` lanes/row
BlockedAdj &gatmh_blocked_for(dory_ctx *c, bool csc, uint32_t ld) {
    if (ld < 128 && (csc ? c->blkIn16_built : c->blkOut16_built)) return csc ? c->blkIn16 : c->blkOut16;
    return csc ? c->blkIn : c->blkOut;
}

int ensure_blocked(dory_ctx *c, bool csc, int group, bool narrow_set) {
    BlockedAdj &B = narrow_set ? (csc ? c->blkIn16 : c->blkOut16) : (csc ? c->blkIn : c->blkOut);
    bool &built = narrow_set ? (csc ? c->blkIn16_built : c->blkOut16_built) : (csc ? c->blkIn_built : c->blkOut_built);
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
        const uint64_t window = 0;   // K1b's own windows (K1s has its own layout: ensure_sweep)
        const uint32_t nb = plan_blocks(NG, want_nb, (uint32_t)group * 16u, window);
        // ... and pointless when the whole source slab fits one XCD's L2 anyway (Cora-sized graphs):
        // K1 then gathers from L2 without partial sums or a second kernel
        // (a partitioned multi-head GAT run has no row-wise form that reads ghost rows: it always takes the blocks)
        const bool tiny = !want_nb && (uint64_t)NG * group * 16u <= ((uint64_t)4 << 20) &&
                          !(c->gnn == DORY_GATMH && c->numNodes > 1);
        if (tiny || nb > 256 || (uint64_t)nb * (c->N + 1) * 8ull > ((uint64_t)8 << 30)) {
            if (!narrow_set) (csc ? c->blkIn_na : c->blkOut_na) = true;   // (no second pair: the narrow layers share the first)
            return DORY_OK;
        }
        HIPCK(c, build_blocked(csc ? c->colPtr : c->rowPtr, csc ? c->rowIdx : c->colIdx, csc ? c->cscVal : c->csrVal,
                               c->N, NG, csc ? c->nnz_in : c->nnz_out, want_nb, (uint32_t)group * 16u, &B, c->compute, window));
        B.row_bytes = (uint32_t)group * 16u;
        built = true;
    }
    return DORY_OK;
}

// K1s bookkeeping: the even layout build_blocked_sweep makes of one adjacency (spmm.hip)
int ensure_sweep(dory_ctx *c, bool csc, int group) {
    BlockedAdj &S = csc ? c->swpIn : c->swpOut;
    bool &built = csc ? c->swpIn_built : c->swpOut_built;
    bool &na = csc ? c->swpIn_na : c->swpOut_na;
    const uint32_t want_nb = (uint32_t)c->opt["spmm_blk_nb"];
    if (built && (csc ? c->swpIn_want_nb : c->swpOut_want_nb) != want_nb && !c->capturing) {   // another block count requested (tests): rebuild
        HIPCK(c, hipStreamSynchronize(c->compute));
        free_blocked(&S);
        built = false;
    }
    if (built || na) return DORY_OK;
    if (c->capturing) return fail(c, DORY_ERR_ARG, "epoch graph: the sweep layout would have to be built while recording");
    const uint32_t NG = c->N + (csc ? c->Gsrc : c->Gdst);
    // the deal is made for the 32-lane launches; the multi-head GAT passes keep ten registers per row: 4 rows at most
    int R;
    if (c->gnn == DORY_GATMH) {
        // four rows while that fills every CU at least once (fewer sweeps = fewer refills of every window: 4.47 -> 4.17 ms per
        // 128-float forward launch from two rows to four); small partitions pick by fill
        const uint32_t G_ = std::min<uint32_t>(32u, c->cus_per_xcd);
        const int forced = (int)c->opt["gatmh_sweep_rows"];
        R = (!forced && c->N >= 8u * G_ * 32u * 4u) ? 4 : sweep_pick_r(c->N, 32, G_, forced, 4);
    } else {
        R = sweep_pick_r(c->N, 32, std::min<uint32_t>(32u, c->cus_per_xcd), (int)c->opt["spmm_sweep_rows"]);
    }
    // source window per block.  0 = by the rows a lane group holds: a step costs ~3 us whatever it gathers, and a small
    // partition (one rank of 8: four rows per group) gathers little per step -- fewer, larger windows win there although
    // two of them no longer fit the L2 (measured, one rank of 8 of the Reddit-size graph: 2432 / 3072 / 3584 / 4096 / 5120 KB
    // = 3.34 / 3.22 / 3.16 / 3.21 / 3.38 ms per epoch; ranks of 4, 2 and the whole graph: 2432 KB stays best)
    // (multi-head GAT contexts: their sweeps carry 13-17 vector instructions per gather and two to four rows per group, and run
    // best on 4.5 MB windows -- 128-float forward 4.45 / 4.16 / 4.09 / 4.29 / 4.88 ms at 2432 / 3584 / 4608 / 6144 / 8192 KB)
    // (round 6: the 8-head GAT's SOURCE side gathers a 128-byte statistics record beside every 512-byte row -- its window is a
    // quarter larger than the forward's for the same rows, and at 4.5 MB of rows it ran fabric-bound: 29.6 GB fetched in 5.1 ms
    // per 128-float launch; the out-edge layout therefore gets its own window, option gatmh_src_window_kb)
    const uint64_t gat_kb = (c->gnn == DORY_GATMH && !csc && c->opt["gatmh_src_window_kb"]) ? (uint64_t)c->opt["gatmh_src_window_kb"] : 4608u;
    const uint64_t window_kb = c->opt["spmm_sweep_window_kb"] ? (uint64_t)c->opt["spmm_sweep_window_kb"]
                                                              : (c->gnn == DORY_GATMH && R >= 4 ? gat_kb : (R <= 4 ? 3584u : 2432u));
    const uint64_t window = window_kb << 10;
    const uint64_t nb_est = ((uint64_t)NG * group * 16u + window - 1) / window + 1;
    // the whole source slab in one L2 (Cora-sized graphs): K1 gathers from L2 anyway.  Thousands of windows (Amazon-,
    // Friendster-sized partitions on a random graph: a row has a fraction of an edge per window): the per-(block,
    // position) offset table alone would be nb*(N+1) words -- K1
    const bool tiny = !want_nb && (uint64_t)NG * group * 16u <= ((uint64_t)4 << 20);
    if (tiny || c->N < 8 || (want_nb ? want_nb : nb_est) > 512 || (want_nb ? want_nb : nb_est) * (uint64_t)(c->N + 1) * 8ull > ((uint64_t)8 << 30)) {
        na = true;
        return DORY_OK;
    }
    HIPCK(c, build_blocked_sweep(csc ? c->colPtr : c->rowPtr, csc ? c->rowIdx : c->colIdx, csc ? c->cscVal : c->csrVal, c->N,
                                 NG, csc ? c->nnz_in : c->nnz_out, want_nb, (uint32_t)group * 16u, window, R, &S, c->compute,
                                 (uint32_t)c->opt["spmm_sweep_layout"], std::min<uint32_t>(32u, c->cus_per_xcd),
                                 group == 32 && c->opt["spmm_sweep_loader"] ? (uint32_t)c->opt["spmm_sweep_loader_relief"] : 0u));
    (csc ? c->swpIn_want_nb : c->swpOut_want_nb) = want_nb;
    built = true;
    return DORY_OK;
}

int blk_group_for(dory_ctx *c, uint32_t ld) {
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
    c->last_spmm_unit = false;
    SpmmArgs a{};
    a.N = c->N; a.F = F; a.ld = xl.ld;
    a.ptr = csc ? c->colPtr : c->rowPtr;
    a.idx = csc ? c->rowIdx : c->colIdx;
    a.val = val;
    a.self_scale = c->norm;
    a.self_mode = self_mode;
    a.xl = xl.d; a.xg = (xg && xg->rows) ? xg->d : nullptr; a.out = out.d;   // nullptr: no ghost rows (the blocked kernel then skips the select)
    a.accumulate = accumulate;
    a.order = (c->opt["spmm_order"] >= 2 || (c->opt["spmm_order"] == 1 && (csc ? c->skewIn : c->skewOut))) ? (csc ? c->orderIn : c->orderOut) : nullptr;
    const bool static_vals = val == (csc ? c->cscVal : c->csrVal) && !(c->gnn == DORY_GAT && csc);  // GAT rewrites cscVal
    if (c->opt["spmm_variant"] == 2 && (static_vals || row_scale) && c->N > 0 && a.ld >= 32) {
        // K1s: register accumulators, every workgroup sweeps all source blocks of its own even layout (spmm.hip).
        const int group = blk_group_for(c, a.ld);
        int rc = ensure_sweep(c, csc, group);
        if (rc) return rc;
        BlockedAdj &S = csc ? c->swpIn : c->swpOut;
        if (!(csc ? c->swpIn_na : c->swpOut_na) && sweep_supported(a, S, group)) {
            const uint32_t G = std::min<uint32_t>(32u, c->cus_per_xcd);
            // With ghost rows the blocks that hold local rows only always run as a launch of their own (they do not
            // depend on an exchange in flight), so the overlapped and the sequential schedule are the same arithmetic.
            const bool two = a.xg != nullptr && S.nb_local > 0 && S.nb_local < S.nb;
            const int force_r = (int)c->opt["spmm_sweep_rows"];
            const size_t need = sweep_scratch_bytes(S, a.ld, group, G, two ? std::max(S.nb_local, S.nb - S.nb_local) : S.nb, force_r);
            if (need > c->partial_bytes) {
                if (c->capturing) return fail(c, DORY_ERR_ARG, "epoch graph: sweep counters would have to grow while recording");
                HIPCK(c, hipStreamSynchronize(c->compute));
                if (c->partial) (void)hipFree(c->partial);
                c->partial = nullptr;
                c->partial_bytes = 0;
                HIPCK(c, hipMalloc((void **)&c->partial, need));
                c->partial_bytes = need;
            }
            if (S.nslots && (rc = ensure_scratch(c, (size_t)S.nslots * a.ld * sizeof(float)))) return rc;   // pieces of split rows
            c->last_spmm_unit = row_scale != nullptr;
            uint32_t *done = reinterpret_cast<uint32_t *>(c->partial);
            uint32_t sflags = (uint32_t)c->opt["spmm_sweep_flags"];
            SweepCtl ctl;
            ctl.force_r = force_r;
            ctl.pair = (int)c->opt["spmm_sweep_pair"];
            ctl.stat = c->sweep_stat;
            ctl.loader = c->opt["spmm_sweep_loader"] != 0;
            SpmmArgs a1 = a;          // the pieces' slots are written, not accumulated, by the first launch
            // placement check failed (ctx.hpp): gates would synchronise workgroups that do not share an L2.  Decide once, by
            // measurement, on a launch that may be repeated (it writes, does not accumulate): gated against ungated.  The
            // three probe launches are timed under their own key ("spmm_xcd_probe"), outside the caller's "spmm" region.
            if ((!c->xcd_mapping_ok || c->opt["spmm_xcd_assume_mismatch"]) && !(sflags & 8u)) {
                if (c->xcd_policy < 0 && !c->capturing && !a.accumulate && !c->halo_pending) {
                    struct Ev3 {   // destroyed on every way out (HIPCK returns)
                        hipEvent_t e[3] = {nullptr, nullptr, nullptr};
                        ~Ev3() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
                    } ev;
                    Timed tp(c, "spmm_xcd_probe", c->compute);
                    for (auto &x : ev.e) HIPCK(c, hipEventCreate(&x));
                    const uint32_t hi = two ? S.nb_local : S.nb;
                    HIPCK(c, launch_spmm_sweep(a1, S, group, row_scale, G, 0, hi, done, c->compute, ctl, sflags | 8u, c->scratch));   // (warm: layout, code)
                    HIPCK(c, hipEventRecord(ev.e[0], c->compute));
                    HIPCK(c, launch_spmm_sweep(a1, S, group, row_scale, G, 0, hi, done, c->compute, ctl, sflags, c->scratch));
                    HIPCK(c, hipEventRecord(ev.e[1], c->compute));
                    HIPCK(c, launch_spmm_sweep(a1, S, group, row_scale, G, 0, hi, done, c->compute, ctl, sflags | 8u, c->scratch));
                    HIPCK(c, hipEventRecord(ev.e[2], c->compute));
                    HIPCK(c, hipEventSynchronize(ev.e[2]));
                    (void)hipEventElapsedTime(&c->xcd_gated_ms, ev.e[0], ev.e[1]);
                    (void)hipEventElapsedTime(&c->xcd_ungated_ms, ev.e[1], ev.e[2]);
                    c->xcd_policy = c->xcd_gated_ms <= c->xcd_ungated_ms ? 0 : 8;
                }
                sflags |= c->xcd_policy == 0 ? 0u : 8u;      // undecided (recording, accumulating caller): ungated, never a timeout
            }
            Timed t(c, "spmm", c->compute);
            if (two) {
                // under an exchange in flight the RCCL kernels need CUs of their own
                const uint32_t reserve = c->halo_pending ? (uint32_t)c->opt["spmm_sweep_reserve_cus"] : 0u;
                {
                    Timed tb(c, c->halo_pending ? "spmm_beside_halo" : "spmm_local_first", c->compute);
                    HIPCK(c, launch_spmm_sweep(a1, S, group, row_scale, G, 0, S.nb_local, done, c->compute, ctl, sflags, c->scratch, reserve));
                }
                if ((rc = wait_halo(c))) return rc;
                SpmmArgs a2 = a;
                a2.self_mode = 0;
                a2.accumulate = 1;
                HIPCK(c, launch_spmm_sweep(a2, S, group, row_scale, G, S.nb_local, S.nb, done, c->compute, ctl, sflags | 2u, c->scratch));
            } else {
                if ((rc = wait_halo(c))) return rc;
                HIPCK(c, launch_spmm_sweep(a1, S, group, row_scale, G, 0, S.nb, done, c->compute, ctl, sflags, c->scratch));
            }
            HIPCK(c, launch_spmm_sweep_combine(a, S, row_scale, c->scratch, c->compute));
            return DORY_OK;
        }
    }
    if (c->opt["spmm_variant"] >= 1 && (static_vals || row_scale) && c->N > 0 && a.ld >= 32) {
        const int group = blk_group_for(c, a.ld);
        int rc = ensure_blocked(c, csc, group);
        if (rc) return rc;
        BlockedAdj &B = csc ? c->blkIn : c->blkOut;
        const size_t need = blocked_partial_bytes(a, B);
        if (!(csc ? c->blkIn_na : c->blkOut_na) && B.nb > 0 && need <= ((size_t)48 << 30)) {
            c->last_spmm_unit = row_scale != nullptr;
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
                {
                    Timed tb(c, c->halo_pending ? "spmm_beside_halo" : "spmm_local_first", c->compute);
                    HIPCK(c, launch_spmm_blocked_part(a, B, c->partial, group, row_scale != nullptr, 0, nb_local, c->compute));
                }
                if ((rc = wait_halo(c))) return rc;
                HIPCK(c, launch_spmm_blocked_part(a, B, c->partial, group, row_scale != nullptr, nb_local, B.nb, c->compute));
            } else {
                if ((rc = wait_halo(c))) return rc;
                HIPCK(c, launch_spmm_blocked_part(a, B, c->partial, group, row_scale != nullptr, 0, B.nb, c->compute));
            }
            if (B.nchunks) {   // hubs: the remainder of the (block,row) segments K1b stopped in
                if ((rc = ensure_scratch(c, (size_t)B.nchunks * a.ld * sizeof(float)))) return rc;
                HIPCK(c, launch_spmm_blocked_long_segments(a, B, c->partial, row_scale != nullptr, c->scratch, c->compute));
            }
            HIPCK(c, launch_spmm_blocked_reduce(a, B, c->partial, row_scale, c->compute));
            return DORY_OK;
        }
    }
    if (!val) return fail(c, DORY_ERR_ARG, "spmm: no edge values");
    const LongRowsDev &longRows = csc ? c->longIn : c->longOut;
    if (longRows.nchunks) {   // hubs: K1 stops after LONG_ROW_CLAMP edges of a row, workgroup-per-chunk kernels do the rest
        int rc = ensure_scratch(c, (size_t)longRows.nchunks * a.ld * sizeof(float));
        if (rc) return rc;
        a.row_clamp = LONG_ROW_CLAMP;
    }
    // K1 under an exchange in flight.  GCN partitions with ghosts hold a local-first copy of the edges (ctx.hpp: EdgeSplit): one
    // launch sums every row's local-source edges beside the exchange, a second one goes on from those sums with the ghost-
    // source edges (boundary rows only) -- the additions happen in the same order as in ONE launch over the copy, which is what
    // runs when nothing is in flight: overlapped and sequential schedule give the same bits.
    uint32_t *split = csc ? c->splitIn : c->splitOut;
    const uint32_t nInt = csc ? c->nIntIn : c->nIntOut;
    const EdgeSplit &es = csc ? c->esIn : c->esOut;
    // (layer 0's forward aggregation reads ghost rows that came from a file: no exchange ever precedes it, in either schedule,
    // so it keeps the reference's edge order -- the local-first copy costs the 300-float Amazon launch a few per cent)
    if (es.idx && a.xg && val == (csc ? c->cscVal : c->csrVal) && !longRows.nchunks && !a.accumulate && c->opt["spmm_edge_split"] &&
        !c->agg_static_ghosts) {
        a.idx = es.idx;
        a.val = es.val;
        Timed t(c, "spmm", c->compute);
        if (c->halo_pending || c->opt["spmm_blk_force_split"]) {
            SpmmArgs p1 = a;
            p1.ptr_end = es.mid;
            {
                Timed tb(c, c->halo_pending ? "spmm_beside_halo" : "spmm_local_first", c->compute);
                HIPCK(c, launch_spmm(p1, (int)c->opt["spmm_variant"], (int)c->opt["spmm_slab"], c->compute));
            }
            int rc = wait_halo(c);
            if (rc) return rc;
            SpmmArgs p2 = a;
            p2.ptr = es.mid;
            p2.ptr_end = a.ptr + 1;
            p2.self_mode = 0;
            p2.accumulate = 2;
            if (split && nInt < c->N) { p2.order = split + nInt; p2.rows = c->N - nInt; }   // rows without a ghost source are done
            if (!split || nInt < c->N) HIPCK(c, launch_spmm(p2, (int)c->opt["spmm_variant"], (int)c->opt["spmm_slab"], c->compute));
            return DORY_OK;
        }
        int rc = wait_halo(c);
        if (rc) return rc;
        HIPCK(c, launch_spmm(a, (int)c->opt["spmm_variant"], (int)c->opt["spmm_slab"], c->compute));
        return DORY_OK;
    }
    // (other cases -- GAT's per-epoch edge values, hub rows: the rows whose sources are all local run first, the rows that read
    // ghost rows after the comm stream's event)
    const bool split_rows = (c->halo_pending || c->opt["spmm_blk_force_split"]) && split && nInt > 0 && nInt < c->N &&
                            !longRows.nchunks;
    Timed t(c, "spmm", c->compute);
    if (split_rows) {
        SpmmArgs part = a;
        part.order = split;
        part.rows = nInt;
        {
            Timed tb(c, c->halo_pending ? "spmm_beside_halo" : "spmm_local_first", c->compute);
            HIPCK(c, launch_spmm(part, (int)c->opt["spmm_variant"], (int)c->opt["spmm_slab"], c->compute));
        }
        int rc = wait_halo(c);
        if (rc) return rc;
        part.order = split + nInt;
        part.rows = c->N - nInt;
        HIPCK(c, launch_spmm(part, (int)c->opt["spmm_variant"], (int)c->opt["spmm_slab"], c->compute));
        return DORY_OK;
    }
    {
        int rc = wait_halo(c);
        if (rc) return rc;
    }
    HIPCK(c, launch_spmm(a, (int)c->opt["spmm_variant"], (int)c->opt["spmm_slab"], c->compute));
    if (longRows.nchunks) HIPCK(c, launch_spmm_long_rows(a, longRows, c->scratch, c->compute));
    return DORY_OK;
}


}  // namespace dory

using namespace dory;

extern "C" {

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
            if (tf_layer(c, layer)) {   // z_l = A (in_l W_l): gather d[l+1]-wide
                NEED(xw, layer, "xw"); NEED(fgxw, layer, "fgxw"); NEED(z, layer, "z");
                if (layer == 0) {   // the input and its ghost rows are static: transform both here
                    Tensor &W = c->weights[0]["w"];
                    int rc = gemm(c, 0, 0, c->N, c->dims[1], c->dims[0], *in, W, *xw);
                    if (!rc && c->Gsrc) rc = gemm(c, 0, 0, c->Gsrc, c->dims[1], c->dims[0], *fg, W, *fgxw);
                    if (rc) return rc;
                }   // deeper layers: apply_vertex(l-1) left xw@l, the forward exchange of layer l its ghost rows
                return spmm(c, true, c->cscVal, 1, *xw, fgxw, *z, c->dims[layer + 1], 0);
            }
            // Opt-in "gcn_cache_ah0" (no reference counterpart; the reference recomputes it every epoch and so does the
            // default here): in full-graph training ah@0 = A_hat [x ; fg@0] is a constant of the run -- x and fg@0 come
            // from files, the adjacency never changes -- and with 288 GB of HBM it can simply stay.  The aggregation of
            // layer 0 is skipped while nothing it reads has been written through this ABI since it was last computed.
            if (layer == 0 && c->opt["gcn_cache_ah0"] && !c->capturing) {
                if (c->ah0_valid) { c->ah0_skips++; return DORY_OK; }
                c->agg_static_ghosts = true;
                int rc = spmm(c, true, c->cscVal, 1, *in, fg, *ah, c->dims[layer], 0);
                c->agg_static_ghosts = false;
                c->ah0_valid = rc == DORY_OK;
                return rc;
            }
            if (layer == 0) c->ah0_valid = false;
            c->agg_static_ghosts = layer == 0;
            const int src_ = spmm(c, true, c->cscVal, 1, *in, fg, *ah, c->dims[layer], 0);
            c->agg_static_ghosts = false;
            return src_;
        }
        if (tf_layer(c, layer)) {   // u_l = A^T g_l (ghost rows of g_l: backward exchange of layer l); dW_l = in_l^T u_l
            NEED(g, layer, "g"); NEED(bgg, layer, "bgg"); NEED(u, layer, "u");
            Tensor *in = layer == 0 ? find(c, 0, "x") : find(c, layer - 1, "h");
            if (!in) return fail(c, DORY_ERR_ARG, "aggregate: input tensor missing");
            const uint32_t Fin = c->dims[layer], Fout = c->dims[layer + 1];
            int rc = spmm(c, false, c->csrVal, 1, *g, bgg, *u, Fout, 0);
            if (rc) return rc;
            if ((rc = gemm(c, 1, 0, Fin, Fout, c->N, *in, *u, c->wgrads[layer]["w"]))) return rc;
            if (layer == 0) return DORY_OK;
            NEED(aTg, layer - 1, "aTg");   // the gradient handed down: A^T (g_l W_l^T) = u_l W_l^T
            return gemm(c, 0, 1, c->N, Fin, Fout, *u, c->weights[layer]["w"], *aTg);
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
                const BlockedAdj &Bf = gatmh_blocked_for(c, true, z->ld);
                const bool blocked = c->opt["gatmh_blocked"] && c->blkIn_built && !c->blkIn_na && Bf.nb > 0 &&
                                     (D % 4 == 0 || K == 1) &&
                                     (size_t)Bf.nb * c->N * z->ld * sizeof(float) <= c->partial_bytes;
                NEED(fgz, fl, "fg_z"); NEED(fgel, fl, "fg_el");
                const BlockedAdj &Sf = c->swpIn;
                const int shl = gatmh_sweep_hl(K, D, z->ld);
                Tensor *op = find(c, fl, "op"), *dpos = find(c, fl, "dpos");
                // (the same addressing test the launchers make -- 32-bit byte offsets through the buffer resource -- so that a
                // partition they would refuse takes the blocked kernels instead of failing, as spmm() does)
                SpmmArgs sa{};
                sa.N = c->N; sa.ld = z->ld;
                const bool sweep = c->opt["gatmh_sweep"] && c->opt["spmm_variant"] == 2 && c->swpIn_built && !c->swpIn_na && Sf.nb > 0 &&
                                   shl != 0 && op && dpos && sweep_supported(sa, Sf, z->ld >= 128 ? 32 : 16);
                if (fl < c->gatmh_fwd_swept.size()) c->gatmh_fwd_swept[fl] = 0;
                if (sweep) {
                    // K1s's skeleton: sums in registers over all source blocks, single-pass softmax against the upper-bound shift
                    int src_ = ensure_scratch(c, gatmh_sweep_scratch_bytes(Sf, c->N, z->ld, el->ld));
                    if (src_) return src_;
                    const uint32_t G = std::min<uint32_t>(32u, c->cus_per_xcd);
                    const bool two = c->Gsrc > 0 && Sf.nb_local > 0 && Sf.nb_local < Sf.nb;
                    const int sgroup = z->ld >= 128 ? 32 : 16;
                    const size_t need = sweep_scratch_bytes(Sf, z->ld, sgroup, G, two ? std::max(Sf.nb_local, Sf.nb - Sf.nb_local) : Sf.nb, gatmh_sweep_rows(Sf, sgroup, shl, 0));
                    if (need > c->partial_bytes) return fail(c, DORY_ERR_ARG, "multi-head GAT sweep: gate counters not allocated (preallocate)");
                    SweepCtl ctl;
                    ctl.stat = c->sweep_stat;
                    uint32_t *done = reinterpret_cast<uint32_t *>(c->partial);
                    const uint32_t sflags = (uint32_t)c->opt["spmm_sweep_flags"] |
                                            (((!c->xcd_mapping_ok || c->opt["spmm_xcd_assume_mismatch"]) && c->xcd_policy != 0) ? 8u : 0u);
                    const float *a_l = c->weights[fl]["a_l"].d;
                    HIPCK(c, launch_gatmh_sweep_begin(c->N, c->Gsrc, K, z->ld, el->ld, Sf, el->d, fgel->d, c->scratch, c->compute));
                    if (two) {
                        HIPCK(c, launch_gatmh_forward_sweep_part(c->N, K, D, z->ld, el->ld, Sf, z->d, fgz->d, er->d, a_l, o->d, op->d, c->scratch, G, 0,
                                                                 Sf.nb_local, false, done, ctl, sflags, c->compute, el->d, fgel->d));
                        if ((src_ = wait_halo(c))) return src_;
                        HIPCK(c, launch_gatmh_forward_sweep_part(c->N, K, D, z->ld, el->ld, Sf, z->d, fgz->d, er->d, a_l, o->d, op->d, c->scratch, G,
                                                                 Sf.nb_local, Sf.nb, true, done, ctl, sflags, c->compute, el->d, fgel->d));
                    } else {
                        if ((src_ = wait_halo(c))) return src_;
                        HIPCK(c, launch_gatmh_forward_sweep_part(c->N, K, D, z->ld, el->ld, Sf, z->d, c->Gsrc ? fgz->d : nullptr, er->d, a_l, o->d,
                                                                 op->d, c->scratch, G, 0, Sf.nb, false, done, ctl, sflags, c->compute, el->d, fgel->d));
                    }
                    HIPCK(c, launch_gatmh_forward_sweep_finish(c->N, K, D, z->ld, el->ld, c->colPtr, c->rowIdx, Sf, z->d, fgz->d, el->d, fgel->d,
                                                               er->d, o->d, op->d, m->d, den->d, dpos->d, c->scratch, c->compute));
                    if (fl < c->gatmh_fwd_swept.size()) c->gatmh_fwd_swept[fl] = 1;
                }
                else if (blocked) {
                    // "gatmh_fused_stats" (default 1): the blocks' own online softmax + a merge in the reduce kernel instead
                    // of a statistics pass over all edges first; the blocks' (m_b, den_b) live in the scratch buffer
                    float *stat_partial = nullptr;
                    if (c->opt["gatmh_fused_stats"]) {
                        int src_ = ensure_scratch(c, (size_t)2 * Bf.nb * c->N * el->ld * sizeof(float));
                        if (src_) return src_;
                        stat_partial = c->scratch;
                    }
                    HIPCK(c, launch_gatmh_forward_blocked(c->N, K, D, z->ld, el->ld, c->colPtr, c->rowIdx, Bf, z->d,
                                                          fgz->d, el->d, fgel->d, er->d, o->d, m->d, den->d, c->partial,
                                                          c->Gsrc > 0, c->compute, stat_partial,
                                                          c->opt["gatmh_el_on_the_fly"] ? c->weights[fl]["a_l"].d : nullptr));
                }
                else if (c->numNodes > 1)
                    return fail(c, DORY_ERR_ARG, "multi-head GAT: a partitioned run needs the source-blocked kernels (gatmh_blocked = 1, K*D a shape they cover)");
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
        NEED(st, fl, "st"); NEED(fgz, fl, "fg_z"); NEED(fgel, fl, "fg_el"); NEED(bgdo, fl, "bg_do"); NEED(bgst, fl, "bg_st");
        const int64_t phase = c->opt["gatmh_bwd_phase"];   // 0: whole sweep; 1 / 2: one phase (caller moves the ghost rows)
        if (phase != 2) {
            Timed t(c, "loss", c->compute);
            if (last) {
                NEED(gr, fl, "grad");
                HIPCK(c, launch_gatmh_head_expand(c->N, K, gr->cols, gr->d, gr->ld, dO->d, dO->ld, c->compute));
            } else {
                NEED(dh, fl + 1, "dh");
                HIPCK(c, launch_gatmh_elu_bwd(c->N, o->cols, dh->d, dh->ld, o->d, o->ld, dO->d, dO->ld, c->compute));
            }
        }
        int rc = ensure_scratch(c, (size_t)2048 * z->cols * sizeof(float) + (size_t)c->N * K * 16 + 256);
        if (rc) return rc;
        // The sweep forms (gat_mh_sweep.hip).  Destination side: when this layer's forward ran on the skeleton it left the
        // positive-branch sums, and t / der / st come from a row-wise kernel -- no edge pass.  Source side: the sweep over the
        // out-edges' layout.  Either falls back to the blocked kernels on its own (same m / den / st semantics).
        const int shl = gatmh_sweep_hl(K, D, z->ld);
        Tensor *op = find(c, fl, "op"), *dpos = find(c, fl, "dpos");
        const bool dst_rowwise = c->opt["gatmh_sweep"] && shl && op && dpos && fl < c->gatmh_fwd_swept.size() && c->gatmh_fwd_swept[fl] &&
                                 ((z->ld >> 2) % (uint32_t)shl) == 0;
        const BlockedAdj &So = c->swpOut;
        SpmmArgs sa{};     // the launchers' addressing tests (rows and the 16-byte statistics records through buffer resources): a
        sa.N = c->N; sa.ld = z->ld;   // partition they would refuse takes the blocked kernels
        const bool src_sweep = c->opt["gatmh_sweep"] && c->opt["spmm_variant"] == 2 && shl && c->swpOut_built && !c->swpOut_na && So.nb > 0 &&
                               ((z->ld >> 2) % (uint32_t)shl) == 0 && sweep_supported(sa, So, z->ld >= 128 ? 32 : 16) &&
                               (uint64_t)std::max(c->N, c->Gdst) * K * 16u < (1ull << 32) && K * 16u < (1u << 24);
        if (dst_rowwise && src_sweep) {
            float4 *st4 = reinterpret_cast<float4 *>(st->d);
            const uint32_t lds4 = st->ld / 4;
            if (phase != 2) {
                Timed t(c, "loss", c->compute);
                HIPCK(c, launch_gatmh_dst_rowwise(c->N, K, D, z->ld, el->ld, dO->d, o->d, op->d, dpos->d, er->d, m->d, den->d, tt->d, der->d,
                                                  st4, lds4, c->compute));
            }
            if (phase == 1) return DORY_OK;
            if (phase == 0 && c->numNodes > 1) {   // ghost destinations of the out-edges: their dO and st rows
                if ((rc = exchange_rows(c, DORY_BACKWARD, dO, bgdo, false))) return rc;
                if ((rc = exchange_rows(c, DORY_BACKWARD, st, bgst, false))) return rc;
            }
            if ((rc = ensure_scratch(c, gatmh_src_sweep_scratch_bytes(So, c->N, c->Gdst, K, z->ld, el->ld)))) return rc;
            const uint32_t G = std::min<uint32_t>(32u, c->cus_per_xcd);
            const bool two = c->Gdst > 0 && So.nb_local > 0 && So.nb_local < So.nb;
            const int sgroup = z->ld >= 128 ? 32 : 16;
            const size_t need = sweep_scratch_bytes(So, z->ld, sgroup, G, two ? std::max(So.nb_local, So.nb - So.nb_local) : So.nb, gatmh_sweep_rows(So, sgroup, shl, 1));
            if (need > c->partial_bytes) return fail(c, DORY_ERR_ARG, "multi-head GAT sweep: gate counters not allocated (preallocate)");
            SweepCtl ctl;
            ctl.stat = c->sweep_stat;
            uint32_t *done = reinterpret_cast<uint32_t *>(c->partial);
            const uint32_t sflags = (uint32_t)c->opt["spmm_sweep_flags"] |
                                    (((!c->xcd_mapping_ok || c->opt["spmm_xcd_assume_mismatch"]) && c->xcd_policy != 0) ? 8u : 0u);
            {
                Timed t(c, "spmm", c->compute);
                HIPCK(c, launch_gatmh_src_sweep_begin(c->N, c->Gdst, K, z->ld, el->ld, So, st4, reinterpret_cast<const float4 *>(bgst->d), lds4,
                                                      c->scratch, c->compute));
                if (two) {
                    HIPCK(c, launch_gatmh_src_sweep_part(c->N, c->Gdst, K, D, z->ld, el->ld, So, dO->d, bgdo->d, el->d, dz->d, c->scratch, G, 0,
                                                         So.nb_local, false, done, ctl, sflags, c->compute));
                    HIPCK(c, launch_gatmh_src_sweep_part(c->N, c->Gdst, K, D, z->ld, el->ld, So, dO->d, bgdo->d, el->d, dz->d, c->scratch, G,
                                                         So.nb_local, So.nb, true, done, ctl, sflags, c->compute));
                } else {
                    HIPCK(c, launch_gatmh_src_sweep_part(c->N, c->Gdst, K, D, z->ld, el->ld, So, dO->d, c->Gdst ? bgdo->d : nullptr, el->d, dz->d,
                                                         c->scratch, G, 0, So.nb, false, done, ctl, sflags, c->compute));
                }
                HIPCK(c, launch_gatmh_src_sweep_finish(c->N, K, D, z->ld, el->ld, So, z->d, el->d, dO->d, der->d, c->weights[fl]["a_l"].d,
                                                       c->weights[fl]["a_r"].d, del->d, dz->d, c->scratch, c->compute));
            }
            // (the attention gradients' column sums take the scratch buffer next: the sweep's sums are consumed by then)
            if ((rc = ensure_scratch(c, (size_t)2048 * z->cols * sizeof(float) + 256))) return rc;
            HIPCK(c, launch_gatmh_dattn(c->N, K, D, z->ld, el->ld, z->d, del->d, der->d, c->wgrads[fl]["a_l"].d,
                                        c->wgrads[fl]["a_r"].d, c->scratch, c->scratch_bytes, c->compute));
            return DORY_OK;
        }
        const BlockedAdj &Bbi = gatmh_blocked_for(c, true, z->ld), &Bbo = gatmh_blocked_for(c, false, z->ld);
        const uint32_t nbmax = std::max(Bbi.nb, Bbo.nb);
        if (c->opt["gatmh_blocked"] && c->blkIn_built && c->blkOut_built && !c->blkIn_na && !c->blkOut_na && nbmax > 0 &&
            gatmh_backward_blocked_ok(K, D, z->ld) &&
            (size_t)nbmax * c->N * (z->ld + K) * sizeof(float) <= c->partial_bytes) {
            float4 *st4 = reinterpret_cast<float4 *>(st->d);
            const uint32_t lds4 = st->ld / 4;
            if (phase != 2) {   // destination side: t, der, st
                Timed t(c, "spmm", c->compute);
                HIPCK(c, launch_gatmh_backward_blocked_dst(c->N, K, D, z->ld, el->ld, Bbi, z->d, fgz->d, el->d, fgel->d,
                                                           er->d, m->d, den->d, dO->d, tt->d, der->d, c->partial, st4, lds4,
                                                           c->Gsrc > 0, c->compute,
                                                           // (el from the gathered row only where the forward formed its statistics that
                                                           //  way too: its ELFLY form needs the fused statistics -- same rounding of alpha)
                                                           (c->opt["gatmh_el_on_the_fly"] && c->opt["gatmh_fused_stats"]) ? c->weights[fl]["a_l"].d : nullptr));
            }
            if (phase == 1) return DORY_OK;
            if (phase == 0 && c->numNodes > 1) {   // ghost destinations of the out-edges: their dO and st rows
                if ((rc = exchange_rows(c, DORY_BACKWARD, dO, bgdo, false))) return rc;
                if ((rc = exchange_rows(c, DORY_BACKWARD, st, bgst, false))) return rc;
            }
            Timed t(c, "spmm", c->compute);
            HIPCK(c, launch_gatmh_backward_blocked_src(c->N, K, D, z->ld, el->ld, Bbo, z->d, el->d, dO->d, bgdo->d, st4,
                                                       reinterpret_cast<const float4 *>(bgst->d), lds4, der->d,
                                                       c->weights[fl]["a_l"].d, c->weights[fl]["a_r"].d, del->d, dz->d,
                                                       c->partial, c->Gdst > 0, c->compute));
            HIPCK(c, launch_gatmh_dattn(c->N, K, D, z->ld, el->ld, z->d, del->d, der->d, c->wgrads[fl]["a_l"].d,
                                        c->wgrads[fl]["a_r"].d, c->scratch, c->scratch_bytes, c->compute));
            return DORY_OK;
        }
        if (c->numNodes > 1)
            return fail(c, DORY_ERR_ARG, "multi-head GAT: a partitioned run needs the source-blocked kernels (gatmh_blocked = 1, K*D a shape they cover)");
        Timed t(c, "spmm", c->compute);
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
    // Round 6: with scores that depend on the destination only, BOTH in-edge aggregations of a layer -- the forward's
    // ah = z + arow (.) S and the backward's aTg += drow (.) S -- are row-scaled copies of ONE unweighted neighbour sum
    // S[v] = sum over in-edges of z[src] (ghosts included).  The forward computes S (tensor "nsum", same K1s launch(es), unit
    // weights), the backward reuses it: two aggregations per layer and epoch instead of three ("gat_reuse_nsum"; a caller who
    // replaced z / fg_z / "A" / "dA" in between gets the general path).
    Tensor *nsum = find(c, fl, "nsum"), *ones = find(c, 0, "ones");
    const bool reuse = c->opt["gat_reuse_nsum"] && nsum && ones && fl < c->gat_nsum_valid.size() && z->ld == nsum->ld;
    if (dir == DORY_FORWARD) {
        NEED(ah, fl, "ah");
        Tensor *arow = find(c, fl, "arow");
        const bool fast = arow && fl < c->gat_arow_valid.size() && c->gat_arow_valid[fl];
        if (fl < c->gat_nsum_valid.size()) c->gat_nsum_valid[fl] = 0;
        if (fast && reuse && c->N) {
            if (!c->gat_ones_set) {
                HIPCK(c, hipMemsetD32Async((hipDeviceptr_t)ones->d, 0x3f800000, c->N, c->compute));
                c->gat_ones_set = true;
            }
            int rc = spmm(c, true, c->cscVal, 0, *z, fgz, *nsum, c->dims[layer], 0, ones->d);
            if (rc) return rc;
            if (c->last_spmm_unit) {      // (K1 -- graphs without a blocked layout -- gathers with the per-edge values: not a unit sum)
                Timed t(c, "spmm", c->compute);
                HIPCK(c, launch_row_axpy(ah->d, nsum->d, arow->d, z->d, c->N, z->ld, c->compute));
                c->gat_nsum_valid[fl] = 1;
                return DORY_OK;
            }
        }
        { int mrc = gat_materialize(c, fl, 2); if (mrc) return mrc; }   // (K1 reads the per-edge values)
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
    if (fast && reuse && c->gat_nsum_valid[fl]) {
        Timed t(c, "spmm", c->compute);
        HIPCK(c, launch_row_axpy(aTg->d, nsum->d, drow->d, nullptr, c->N, aTg->ld, c->compute));
        return DORY_OK;
    }
    { int mrc = gat_materialize(c, fl, 4); if (mrc) return mrc; }
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
                if (tf_layer(c, layer)) {   // z_l came out of dory_aggregate already
                    Timed t(c, "loss", c->compute);
                    HIPCK(c, launch_tanh_forward(N, Fout, z->d, z->ld, h->d, h->ld, c->compute));
                } else if ((rc = gemm(c, 0, 0, N, Fout, Fin, *ah, W, *z, h))) {
                    return rc;
                }
                if (tf_layer(c, layer + 1)) {   // the next layer gathers (h_l W_{l+1}): transform before the exchange
                    NEED(xwn, layer + 1, "xw");
                    return gemm(c, 0, 0, N, c->dims[layer + 2], Fout, *h, c->weights[layer + 1]["w"], *xwn);
                }
                return DORY_OK;
            }
            // last layer (CPU_comm.cpp:108-133)
            NEED(lab, layer, "lab");
            const bool tfl = tf_layer(c, layer);   // then z_l came out of dory_aggregate, and dW_l / the handed-down gradient follow there
            if (!tfl && (rc = gemm(c, 0, 0, N, Fout, Fin, *ah, W, *z))) return rc;
            const uint32_t stt = (uint32_t)(N * 0.66);            // TRAIN_PORTION
            const uint32_t vend = stt + (uint32_t)(N * 0.1);      // VAL_PORTION
            c->val_rows = vend - stt;
            const float denom = (float)(c->globalV * 0.66);
            if ((rc = ensure_scratch(c, softmax_xent_scratch_bytes(Fout, vend - stt)))) return rc;
            {
                Timed t(c, "loss", c->compute);
                // maskout copies (N - stt) floats starting at dense offset stt*cols (CPU_comm.cpp:464-471)
                HIPCK(c, launch_softmax_xent(N, Fout, z->d, z->ld, lab->d, lab->ld, g->d, g->ld, denom, stt,
                                             vend, (uint64_t)stt * Fout, (uint64_t)(N - stt), c->d_stat,
                                             c->scratch, c->compute));
            }
            if (tfl) return DORY_OK;
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
        if (tf_layer(c, layer)) return DORY_OK;   // dW_l (and the gradient for layer l-1) follow in dory_aggregate(l, backward)
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
        if (layer < c->gat_nsum_valid.size()) c->gat_nsum_valid[layer] = 0;   // a new z: the kept neighbour sum of the old one no longer holds
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
        if (c->Gsrc) {   // scores of the ghost sources from their exchanged z rows (el is all the in-edge side needs)
            NEED(fgz, l0, "fg_z"); NEED(fgel, l0, "fg_el"); NEED(fger, l0, "fg_er");
            HIPCK(c, launch_gatmh_scores(c->Gsrc, K, z->cols / K, fgz->d, fgz->ld, c->weights[l0]["a_l"].d,
                                         c->weights[l0]["a_r"].d, fgel->d, fger->d, fgel->ld, c->compute));
        }
        return DORY_OK;
    }
    const uint32_t fl = layer - 1;  // "layer--; // YIFAN: fix this" (CPU_comm.cpp:33)
    const uint32_t F = c->dims[fl + 1];
    Tensor &a = c->weights[fl]["a_i"];
    NEED(z, fl, "z");
    NEED(az, fl, "az");
    if (dir == DORY_FORWARD) {  // edgNNForwardGAT (CPU_comm.cpp:190-203)
        NEED(arow, fl, "arow");
        NEED(azrow, fl, "azrow");
        Timed t(c, "edge", c->compute);
        const bool lazy = c->opt["gat_lazy_edge_tensors"] != 0;
        HIPCK(c, launch_edge_forward_gat(c->N, F, c->colPtr, z->d, z->ld, a.d, lazy ? nullptr : az->d, lazy ? nullptr : c->cscVal, arow->d,
                                         c->compute, azrow->d));
        for (auto &f : c->gat_arow_valid) f = 0;   // "A" now holds this layer's scores only
        c->gat_arow_valid[fl] = 1;
        c->gat_azrow_valid[fl] = 1;
        c->gat_az_stale[fl] = lazy ? 1 : 0;        // (lazy: the per-edge copies follow when somebody reads them, gat_materialize)
        c->gat_A_stale_layer = lazy ? (int)fl : -1;
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
    {
        Tensor *azrow = find(c, fl, "azrow");
        const bool have_row = azrow && c->gat_azrow_valid[fl];
        const bool lazy = c->opt["gat_lazy_edge_tensors"] != 0 && have_row;
        if (!have_row) { int mrc = gat_materialize(c, fl, 1); if (mrc) return mrc; }   // (az comes from the caller, or is current already)
        HIPCK(c, launch_edge_backward_gat(c->N, F, c->colPtr, grad->d, grad->ld, az->d, a.d, lazy ? nullptr : dA->d, cw->d, drow->d, c->compute,
                                          have_row ? azrow->d : nullptr));
        c->gat_dA_stale[fl] = lazy ? 1 : 0;
    }
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

}  // extern "C"
