/*
 * dorylus_hip.h -- C-ABI of the MI355X-native aggregation + transform engine.
 *
 * This is the drop-in boundary for the reference's "cpu"/"gpu" backends
 * (uclasystem/dorylus, src/graph-server).  Every entry point replaces one piece
 * of the reference's Engine / ResourceComm interface; the replaced code is cited
 * as file:line relative to src/graph-server/ (or src/ where noted).
 *
 * Conventions
 *   - plain C, no C++/torch types; all functions return 0 on success or a
 *     negative dory_status; dory_last_error() gives the message.  Nothing in the
 *     library aborts the host process (the reference assert()/exit()s,
 *     GPU-Computation/cu_matrix.cu:137-141).
 *   - one dory_ctx per GPU / per graph partition ("node" in the reference).
 *     Calls on one ctx are serialised internally (mutex), so they may come from
 *     the Engine's different stage threads (engine/engine.cpp:243-273).
 *   - all tensors are fp32 (FeatType/EdgeType = float, common/utils.hpp:28-29);
 *     host buffers are dense row-major rows x cols exactly as the Engine's
 *     savedNNTensors (engine/ops/gcn_ops.cpp:27-93); the device copy is
 *     authoritative between dory_tensor_upload and dory_tensor_download.
 *   - layer / dir arguments have the meaning of Chunk::layer / Chunk::dir
 *     (common/utils.hpp:64-75) at the call site they replace.
 */
#ifndef DORYLUS_HIP_H
#define DORYLUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dory_ctx dory_ctx;

enum dory_status {
    DORY_OK = 0,
    DORY_ERR_ARG = -1,     /* bad argument / unknown tensor / wrong state        */
    DORY_ERR_HIP = -2,     /* a HIP runtime call failed                          */
    DORY_ERR_NODEVICE = -3,/* no usable MI355X (gfx950) device                   */
    DORY_ERR_COMM = -4,    /* RCCL failure                                       */
    DORY_ERR_IO = -5       /* file format / IO problem (host-side helpers)       */
};

enum dory_dir { DORY_FORWARD = 0, DORY_BACKWARD = 1 }; /* PROP_TYPE, common/utils.hpp:46 */
enum dory_gnn { DORY_GCN = 0, DORY_GAT = 1,            /* GNN,       common/utils.hpp:48 */
                DORY_GATMH = 2 };  /* extension, not in the reference: multi-head GAT with per-edge
                                      attention softmax (BASELINE config 3 wording) */

/* ---- lifetime ---------------------------------------------------------------
 * Replaces ComputingUnit::getInstance / ComputingServer construction
 * (GPU-Computation/comp_unit.cu:23-46, engine/engine.cpp:141-163). */
int dory_create(int device, dory_ctx **out);
int dory_destroy(dory_ctx *ctx);
const char *dory_last_error(dory_ctx *ctx); /* ctx may be NULL: last create error */

/* Adopt caller-owned HIP streams (hipStream_t passed as void*); NULL keeps the
 * context's own.  compute: all kernels; comm: RCCL + pack/unpack. */
int dory_set_streams(dory_ctx *ctx, void *compute_stream, void *comm_stream);
int dory_sync(dory_ctx *ctx); /* wait for both streams (NodeManager::barrier's local half) */

/* ---- model / partition description -------------------------------------------
 * Replaces Engine::readLayerConfigFile + the Engine fields the backend reads
 * (engine/utils.cpp:460-479; gnn_type, numLayers, layerConfig, nodeId, numNodes,
 * graph.globalVtxCnt).  dims has num_layers+1 entries (run/reddit.config = 602 128 41). */
int dory_configure(dory_ctx *ctx, int gnn_type, uint32_t num_layers,
                   const uint32_t *dims, uint32_t global_vtx_cnt,
                   uint32_t node_id, uint32_t num_nodes);

/* Multi-head GAT extension only (gnn_type == DORY_GATMH): heads per layer (num_layers
 * entries; default 8 for hidden layers, 1 for the last).  Hidden layer l concatenates its
 * heads (dims[l+1] = K*D, D a power of two <= 64, K*D <= 256) and applies ELU; the last layer
 * averages its heads into dims[L] logits.  Call between dory_configure and dory_preallocate.
 * Tensors: h z el er m den o do dz t del der (+ logits grad lab at L-1), weights w a_l a_r.
 * Stage mapping: apply_vertex fwd = z=h*W; apply_edge fwd = el,er; aggregate fwd = edge softmax
 * + weighted sum (+ELU / head mean); predict_gat = softmax - label; aggregate bwd = attention
 * backward (dz, da_l, da_r); apply_vertex bwd = dW, dh.  Partitioned runs: dory_halo_exchange(layer, FORWARD) ships
 * "z" -> "fg_z" (the ghost sources' scores are recomputed from it); the backward sweep of dory_aggregate runs in two
 * phases and ships "do" -> "bg_do" and "st" -> "bg_st" (the packed (er, m, 1/den, t) rows) of the out-edges' ghost
 * destinations in between -- itself over RCCL, or, with option "gatmh_bwd_phase" = 1 / 2, one phase per call so that
 * the caller can move those rows (dory_halo_pack_tensor / dory_halo_unpack_tensor); dory_halo_exchange(layer,
 * BACKWARD) is a no-op for this model.  Partitioned runs need the source-blocked kernels ("gatmh_blocked" = 1).
 * Sweep forms (option "gatmh_sweep" = 1, default; round 5): where the K1s layouts apply and a head spans 2..16 lanes of a
 * 16-byte-per-lane row slab, the forward runs on K1s's skeleton with a single-pass softmax against an upper-bound shift
 * ("m" then holds that shift, "den" the matching denominator: alpha = exp(s - m) / den as before) and leaves two more
 * tensors, "op" (the part of "o" that came over edges on LeakyReLU's positive branch) and "dpos" (their attention mass);
 * the backward's destination side is then a row-wise kernel (t = <do, o>, der = 0.8 (<do, op> - t dpos)) and its source
 * side a sweep over the out-edges' layout.  Other shapes, or "gatmh_sweep" = 0, keep the blocked kernels; the named
 * tensors and the two-phase protocol are the same either way.  "gatmh_sweep_rows": rows per lane group of the layouts. */
int dory_gatmh_heads(dory_ctx *ctx, const uint32_t *heads);

/* Upload one partition's adjacency, exactly the arrays of Graph
 * (graph/graph.hpp:60-99): forwardAdj (CSC over destination columns, in-edges)
 * and backwardAdj (CSR over source rows, out-edges), 64-bit pointers, 32-bit
 * local indices (ghost ids = N + k), per-edge values, vtxDataVec norms.
 * Replaces the upload in Engine::init (engine/engine.cpp:105-125) and
 * CuMatrix::loadSpCSR/loadSpCSC (GPU-Computation/cu_matrix.cu:36-98), without
 * the 32-bit truncation of cu_matrix.cu:55-58. */
int dory_graph_upload(dory_ctx *ctx, uint32_t local_vtx_cnt,
                      uint32_t src_ghost_cnt, uint32_t dst_ghost_cnt,
                      uint64_t nnz_in, const uint64_t *column_ptrs,
                      const uint32_t *row_idxs, const float *csc_values,
                      uint64_t nnz_out, const uint64_t *row_ptrs,
                      const uint32_t *column_idxs, const float *csr_values,
                      const float *vtx_norms);

/* Allocate the named tensor table on the device: Engine::preallocateGCN /
 * preallocateGAT (engine/ops/gcn_ops.cpp:27-93, gat_ops.cpp:27-115).  GCN names:
 * x fg ah z h lab grad bg aTg; GAT names: h z az fg_z A ah grad dA aTg bg_d lab
 * (SURVEY.md Appendix B).  Requires configure + graph_upload. */
int dory_preallocate(dory_ctx *ctx);

/* ---- named tensors (savedNNTensors[layer][name]) ----------------------------- */
int dory_tensor_info(dory_ctx *ctx, uint32_t layer, const char *name,
                     uint64_t *rows, uint32_t *cols, uint32_t *ld, void **device_ptr);
/* host (dense rows x cols) -> device, and back; both synchronous w.r.t. the
 * compute stream.  Replace the per-op cublasSetMatrix/GetMatrix round trips
 * (cu_matrix.cu:101-120,149,173) with explicit, once-per-run transfers. */
int dory_tensor_upload(dory_ctx *ctx, uint32_t layer, const char *name, const float *host);
int dory_tensor_download(dory_ctx *ctx, uint32_t layer, const char *name, float *host);
/* fill a tensor on the device (synthetic inputs): uniform[lo,hi) from a counter
 * RNG keyed by (seed, global_row * cols + col).  global_row_ids (host, `rows`
 * entries, or NULL for identity) makes a vertex's row independent of the
 * partition that holds it (pass localToGlobalId / the ghost gvids). */
int dory_tensor_fill_uniform(dory_ctx *ctx, uint32_t layer, const char *name,
                             uint64_t seed, float lo, float hi,
                             const uint32_t *global_row_ids);
/* one-hot labels from u32 class ids (Engine::readLabelsFile, engine/utils.cpp:559-596) */
int dory_labels_upload(dory_ctx *ctx, const uint32_t *labels);

/* ---- weights (replaces MessageService weight fetch / push,
 * commmanager/message_service.cpp:142-242, and the weight server's store) ------
 * name is "w" (dims[l] x dims[l+1]) or "a_i" (dims[l+1] x 1, GAT). */
int dory_weight_set(dory_ctx *ctx, uint32_t layer, const char *name, const float *host);
int dory_weight_get(dory_ctx *ctx, uint32_t layer, const char *name, float *host);
int dory_weight_grad_get(dory_ctx *ctx, uint32_t layer, const char *name, float *host);
/* the weight server's side of a push (WeightTensor::localUpdate, weighttensor.cpp:131-150): overwrite the
 * gradient the next dory_weight_update applies -- for callers that sum the updates themselves */
int dory_weight_grad_set(dory_ctx *ctx, uint32_t layer, const char *name, const float *host);
/* WeightServer::xavierInitializer, seed 8888 (src/weight-server/weightserver.cpp:567-585) */
int dory_weights_init_xavier(dory_ctx *ctx);

/* ---- the hot path ---------------------------------------------------------------
 * dory_aggregate      = Engine::aggregateGCN / aggregateGAT (Chunk c) for the whole
 *                       local partition [0, N)  (gcn_ops.cpp:130-191, gat_ops.cpp:173-243)
 * dory_apply_vertex   = ResourceComm::NNCompute(chunk) with chunk.vertex == true
 *                       (CPU_comm.cpp:22-44 -> vtxNNForward / vtxNNBackward, :98-188)
 * dory_apply_edge     = NNCompute with chunk.vertex == false (CPU_comm.cpp:33-42 ->
 *                       edgNNForwardGAT / edgNNBackwardGAT, :190-242); `layer` is the
 *                       chunk's layer, the callee applies the reference's layer-1.
 * dory_predict_gat    = Engine::predictGAT (gat_ops.cpp:246-265)
 */
int dory_aggregate(dory_ctx *ctx, uint32_t layer, int dir);
int dory_apply_vertex(dory_ctx *ctx, uint32_t layer, int dir);
int dory_apply_edge(dory_ctx *ctx, uint32_t layer, int dir);
int dory_predict_gat(dory_ctx *ctx, uint32_t layer);

/* validation accuracy / loss sums of the last forward pass
 * (CPUComm::getTrainStat, CPU_comm.cpp:448-462; MessageService::sendAccloss) */
int dory_train_stat(dory_ctx *ctx, float *acc_sum, float *loss_sum, uint32_t *val_rows);
/* The same summed over all partitions -- what the weight servers do with the AccLoss records the graph servers send them
 * (WeightServer::updateLocalAccLoss / updateGlobalAccLoss, src/weight-server/weightserver.cpp:190-262: vtcsCnt, acc, loss
 * added up over the nodes, "Epoch %u, acc: %.4f, loss: %.4f" on node 0).  A collective: every rank calls it, every rank gets
 * the sums (RCCL all-reduce of three scalars, or the local / host transport).  num_nodes == 1: the local values. */
int dory_train_stat_global(dory_ctx *ctx, float *acc_sum, float *loss_sum, uint32_t *val_rows);

/* ---- ghost-vertex halo exchange (replaces Engine::scatterGCN/GAT +
 * verticesPushOut + ghostReceiver*, gcn_ops.cpp:204-362, engine/utils.cpp:623-650)
 * The plan is the per-peer send lists of graph.<id>.bin (forwardGhostsList /
 * backwardGhostsList, graph/dataloader.cpp:277-297) plus, per peer, the ghost
 * slots its rows land in (srcGhostVtcs / dstGhostVtcs order, dataloader.cpp:311-322).
 * counts arrays have num_nodes entries; lists are concatenated in peer order. */
int dory_halo_plan(dory_ctx *ctx, int dir, const uint32_t *send_counts,
                   const uint32_t *send_lvids, const uint32_t *recv_counts,
                   const uint32_t *recv_slots);
/* RCCL communicator over xGMI: rank 0 makes an id (128 bytes), every rank passes
 * the same bytes.  With num_nodes == 1 none of this is needed. */
int dory_comm_unique_id(void *id128);
int dory_comm_init(dory_ctx *ctx, const void *id128, int rank, int nranks);
/* Host transport instead of RCCL: the exchange step and the gradient sum call back into the host program with
 * host buffers -- the seam where the reference's own CommManager / ZeroMQ data path (gcn_ops.cpp:204-362:
 * verticesPushOut + ghostReceiver*, weighttensor.cpp:131-166 for the update sum) or any MPI can carry the bytes.
 * Everything else of the multi-rank path (plan, pack, unpack, stream ordering, overlap with the local-source
 * aggregation, Adam) stays the library's.  alltoallv: floats, per-peer counts and offsets (num_nodes entries
 * each, self entry 0); allreduce: in-place sum over all ranks.  Callbacks return 0 on success and run on the
 * calling thread, between two stream synchronisations.  Passing NULLs returns to RCCL. */
typedef int (*dory_alltoallv_fn)(void *user, const float *send, const uint64_t *send_counts, const uint64_t *send_offsets,
                                 float *recv, const uint64_t *recv_counts, const uint64_t *recv_offsets, uint32_t num_nodes);
typedef int (*dory_allreduce_fn)(void *user, float *buf, uint64_t n);
int dory_comm_set_host_transport(dory_ctx *ctx, dory_alltoallv_fn alltoallv, dory_allreduce_fn allreduce_sum, void *user);
/* In-process device transport: the `n` contexts (rank i = ctxs[i], all configured with num_nodes == n, graphs and halo
 * plans uploaded, preallocated, on ONE device) become each other's peers.  An exchange is then pack -> hipMemcpyAsync
 * device -> device into every peer's receive buffer on the SENDER's comm stream -> cross-context events -> unpack on the
 * receiver's comm stream; the gradient sum reads the peers' gradients in rank order.  Same stream / event structure as the
 * RCCL path and no host synchronisation with the device: the overlapped schedule of Engine::scatterGCN + ghostReceiver
 * (gcn_ops.cpp:204-282 send, :284-362 receive) with copies that really run beside the aggregation, on one GPU.  Drive
 * every rank from its own host thread (as real ranks are), or stage by stage; a rank whose peer's host thread never
 * arrives fails with DORY_ERR_COMM after option local_timeout_ms (default 30 s).  Destroy the contexts only after all of
 * them are synchronised.  A host transport set on a context takes precedence; RCCL is not used. */
int dory_comm_init_local(dory_ctx *const *ctxs, uint32_t n);
/* pack -> grouped ncclSend/ncclRecv (all-to-all-v) -> unpack into fg / bg
 * (GCN: fwd sends h@(layer-1) into fg@layer, bwd sends grad@layer into bg@(layer-1);
 *  GAT: fwd z@(layer-1) -> fg_z@(layer-1), bwd grad@(layer-1) -> bg_d@(layer-1);
 *  `layer`/`dir` are the chunk's, as in Engine::scatter*).  */
int dory_halo_exchange(dory_ctx *ctx, uint32_t layer, int dir);
/* split entry points for callers that own the transport (tests, other MPI):
 * buffers are device pointers of total_send_rows x cols / total_recv_rows x cols */
int dory_halo_pack(dory_ctx *ctx, uint32_t layer, int dir, float *send_buf);
int dory_halo_unpack(dory_ctx *ctx, uint32_t layer, int dir, const float *recv_buf);
/* The same for any named tensor: rows of a per-local-vertex tensor listed in the plan of `dir` -> send_buf, and
 * recv_buf -> the rows of a ghost tensor of that direction (foreign transports; what the multi-head GAT extension's
 * backward sweep ships between its two phases: "do" -> "bg_do", "st" -> "bg_st"). */
int dory_halo_pack_tensor(dory_ctx *ctx, uint32_t layer, const char *name, int dir, float *send_buf);
int dory_halo_unpack_tensor(dory_ctx *ctx, uint32_t layer, const char *name, int dir, const float *recv_buf);

/* ---- weight-gradient reduction + optimiser (replaces the weight server's
 * PUB/SUB all-gather-and-sum + Adam, src/weight-server/weightserver.cpp:89-187,
 * weighttensor.cpp:131-166,246-328, AdamOptimizer.cpp:29-51) ---------------------
 * dory_weight_update: all-reduce(sum) of the layer's gradient over the
 * communicator (no-op when alone), then one Adam step.  Layer 0 advances the
 * optimiser's iteration count like AdamOptimizer::update. */
int dory_adam_config(dory_ctx *ctx, float learning_rate);
int dory_weight_update(dory_ctx *ctx, uint32_t layer);

/* ---- introspection -------------------------------------------------------------- */
/* what dory_configure / dory_graph_upload recorded (any pointer may be NULL) */
int dory_ctx_describe(dory_ctx *ctx, int *gnn_type, uint32_t *num_layers, uint32_t *node_id,
                      uint32_t *num_nodes, uint32_t *local_vtx_cnt);
/* average device time (ms) and launch count of a kernel family since the last
 * reset, measured with HIP events on the stream the kernel runs on; names:
 * "spmm", "gemm", "loss", "edge", "halo", "allreduce", "adam".  Overlap of an exchange with the aggregation that runs beside
 * it ("halo_overlap"): "halo_deferred" = exchanges the compute stream did not wait for at once, "spmm_beside_halo" = the
 * launches that ran beside them (local-source blocks / interior rows), "halo_hidden" = the part of each such exchange that
 * lies inside its launch's interval on the device clock (total_ms; hidden / deferred = the overlap fraction). */
int dory_timing_enable(dory_ctx *ctx, int on);
int dory_timing_get(dory_ctx *ctx, const char *family, double *total_ms, uint64_t *launches);
int dory_timing_reset(dory_ctx *ctx);
/* tuning knobs (e.g. "spmm_variant", "spmm_slab"); unknown keys are an error.  Read-only keys of dory_get_option:
 * "spmm_gate_timeouts" (K1s sweeps whose workgroups were not co-resident within the polling bound: the launch and the
 * context's next 16 K1s launches ran without gates -- same results, unsynchronised gather rate) and
 * "spmm_ungated_launches"; "epoch_graph_recorded".  dory_timing_get("spmm_gate_timeouts") returns the same pair
 * (launches = timeouts, total_ms = ungated launches).  Write-only key "spmm_gates_rearm": ends a gate back-off at once
 * (a caller that has just changed its cause, e.g. another "spmm_sweep_reserve_cus"); the counters stay (refused while an
 * epoch graph is being recorded; read back: 1 while a back-off is pending).  K1s's placement assumption -- workgroups
 * with equal id & 7 share an XCD, eight XCDs -- is checked once per context at dory_create with HW_REG_XCC_ID probes:
 * read-only "spmm_xcd_mapping_ok", "spmm_xcd_count"; when it does not hold, the first repeatable K1s launch is timed with
 * and without gates and the faster form kept ("spmm_xcd_policy": -1 nothing to decide, 0 gated, 8 ungated;
 * "spmm_xcd_gated_us" / "spmm_xcd_ungated_us"); "spmm_xcd_assume_mismatch" = 1 forces that path (tests).
 * Round 6: "spmm_edge_split" (default 1; before dory_graph_upload): K1 on GCN partitions with ghosts walks a local-first copy
 * of every row's edges, so that an exchange in flight hides under the local-source part of EVERY row; "spmm_sweep_cus"
 * (before dory_graph_upload): workgroups per sweep and XCD of the gated sweeps, for contexts that share a device
 * (dory_comm_init_local); "local_timeout_ms"; "spmm_order" = 3 (before the upload): rows by median source id (experiment);
 * "gatmh_src_window_kb": the 8-head GAT's out-edge sweep layout on its own source window; "gat_reuse_nsum" (default 1): the GAT
 * prototype's backward dA-weighted aggregation from the forward's unweighted neighbour sum (tensor "nsum") instead of a third sweep.  The timing family
 * "spmm_local_first" is the first launch of a two-launch aggregation when no exchange is in flight ("spmm_blk_force_split"). */
int dory_set_option(dory_ctx *ctx, const char *key, int64_t value);
int dory_get_option(dory_ctx *ctx, const char *key, int64_t *value);
/* Diagnostic (no reference counterpart): hold `workgroups` whole CUs for `usec` microseconds with a sleeping kernel on
 * the context's comm stream -- what an exchange's RCCL kernels or a co-tenant's kernels do to the aggregation that runs
 * beside them.  Lets a single-GPU test exercise "spmm_sweep_reserve_cus" and the gate timeouts. */
int dory_debug_occupy_cus(dory_ctx *ctx, uint32_t workgroups, uint64_t usec);

/* Transform-first order of GCN layer 0 (option "gcn_transform_first" = 1; no reference counterpart): when the
 * input is wider than the first hidden layer, z0 = A (X W0) instead of (A X) W0, i.e. the aggregation gathers
 * dims[1]-wide rows, and dW0 = X^T (A^T g0).  Same z0, h0, dW0 within fp32 rounding; "ah"@0 is not produced.
 * Stage meaning in this mode: dory_aggregate(0, FORWARD) writes "z"@0; dory_apply_vertex(0, FORWARD) only applies
 * tanh; dory_apply_vertex(0, BACKWARD) only forms g0; dory_halo_exchange(0, BACKWARD) ships g0's ghost rows;
 * dory_aggregate(0, BACKWARD) forms dW0 -- call dory_weight_update(0) after it.  dory_engine_run follows this
 * order by itself.  Precondition: the backward adjacency values are the forward ones transposed, which holds for
 * partitions built from directed records (the reference datasets are stored symmetrised and run with
 * --undirected 0, run/run-onnode:46); with --undirected 1 the reference counts ghost degrees from file records
 * only (dataloader.cpp:192-218) and the two owners of an edge disagree on its value: dory_partition_upload then
 * sets the option "adjacency_values_asymmetric" and the mode stays off (a direct dory_graph_upload caller sets it
 * itself).  Returns 1 when the mode applies to the configured model and graph, else 0. */
int dory_transform_first_active(dory_ctx *ctx);
/* With "gcn_transform_first" = 2 every layer whose input is wider than its output runs in this order (Reddit: both
 * layers -- aggregations of 128, 41, 41 and 128 floats per edge instead of 602, 128, 128).  For such a layer l > 0:
 * dory_apply_vertex(l-1, FORWARD) also leaves "xw"@l = h_{l-1} W_l, the forward exchange of layer l ships those
 * (narrower) rows, dory_aggregate(l, FORWARD) writes "z"@l, the backward exchange of layer l ships g_l, and
 * dory_aggregate(l, BACKWARD) forms u_l = A^T g_l, dW_l = h_{l-1}^T u_l and "aTg"@(l-1) = u_l W_l^T; the weight
 * update of layer l follows that call.  dory_transform_first_active reports whether any layer is in this mode,
 * dory_transform_first_layer a particular one. */
int dory_transform_first_layer(dory_ctx *ctx, uint32_t layer);

/* Cached layer-0 aggregate (option "gcn_cache_ah0" = 1, default 0; no reference counterpart -- Engine::aggregateGCN
 * recomputes A_hat x every epoch, gcn_ops.cpp:139-148, and so does the default here).  In full-graph training "ah"@0 is a
 * constant of the run: "x" and "fg"@0 come from files and the adjacency never changes.  With the option on,
 * dory_aggregate(0, DORY_FORWARD) returns at once while "ah"@0 still holds the aggregate of the current inputs: it is
 * recomputed after any dory_tensor_upload / dory_tensor_fill_uniform / dory_halo_unpack* of a layer-0 tensor, a
 * dory_halo_exchange(0, DORY_FORWARD) (it rewrites "fg"@0: a peer may hold a new "x"), dory_graph_upload,
 * dory_preallocate or dory_set_option.  (A caller that writes "x" through the device pointer of
 * dory_tensor_info must invalidate by one of those calls itself.)  Results are bit-identical to the uncached run: the
 * same kernel produced the kept tensor.  Not combined with a recorded epoch (the recording always contains the
 * aggregation).  dory_get_option "gcn_cache_ah0_skips" counts the aggregations answered from the kept tensor. */

/* Epoch graph (MI355X-side addition, no reference counterpart): record the calls of one
 * epoch -- dory_aggregate / dory_apply_vertex / dory_apply_edge / dory_predict_gat /
 * dory_weight_update, exactly as Engine::runEpoch issues them -- into a hipGraph and replay
 * it, so a launch-bound epoch costs one graph launch.  Single partition only; one eager epoch
 * must have run before (lazy allocations).  Between begin and end the calls record instead of
 * executing; dory_epoch_graph_launch(n) replays n epochs (asynchronously: dory_sync waits) and
 * advances Adam's iteration count exactly as n eager epochs would (AdamOptimizer.cpp:29-34).
 * dory_engine_run does all of this itself when the option "epoch_graph" is 1. */
int dory_epoch_graph_begin(dory_ctx *ctx);
int dory_epoch_graph_end(dory_ctx *ctx);
int dory_epoch_graph_launch(dory_ctx *ctx, uint32_t epochs);
int dory_epoch_graph_drop(dory_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* DORYLUS_HIP_H */
