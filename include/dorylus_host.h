/*
 * dorylus_host.h -- C-ABI of the host-side pieces either side of the hot path:
 * the partition builder / graph.<id>.bin reader-writer (the data format the
 * kernels consume) and the synchronous-epoch Engine driver.  Same library as
 * dorylus_hip.h (libdorylus_hip.so).  Reference citations are relative to
 * src/graph-server/ of uclasystem/dorylus.
 */
#ifndef DORYLUS_HOST_H
#define DORYLUS_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "dorylus_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- one partition of the graph on the host: every field of graph.<id>.bin ---- */
typedef struct dory_partition dory_partition;

/* DataLoader::preprocess (graph/dataloader.cpp:225-330) on in-memory edge records:
 * local ids ascend with global id, ghost ids are N + rank in ascending global id,
 * edges inside a column/row keep record order (duplicates kept, self loops dropped),
 * edge values (indeg(src)+1)^-1/2 (indeg(dst)+1)^-1/2 computed like
 * setEdgeNormalizations (:153-185), ghost degrees like findGhostDegrees (:192-218).
 * parts[v] = partition of global vertex v.  Index-identical to the reference. */
int dory_partition_build(const uint32_t *src, const uint32_t *dst, uint64_t num_records,
                         const int32_t *parts, uint32_t global_vtx_cnt, uint32_t node_id,
                         uint32_t num_nodes, int undirected, dory_partition **out);
/* same, reading <dataset_dir>graph.bsnap.edges and graph.bsnap.parts
 * (dataloader.cpp:8-10; dataset_dir ends with '/') */
int dory_partition_build_from_files(const char *dataset_dir, uint32_t node_id, uint32_t num_nodes,
                                    int undirected, dory_partition **out);
/* Graph::init (graph/graph.cpp:7-115) / RawGraph::dump (graph.cpp:200-273) */
int dory_partition_load(const char *graph_bin_path, dory_partition **out);
int dory_partition_save(const dory_partition *p, const char *graph_bin_path);
int dory_partition_free(dory_partition *p);
const char *dory_host_last_error(void);

struct dory_partition_view {
    uint32_t local_vtx_cnt, global_vtx_cnt, src_ghost_cnt, dst_ghost_cnt, num_nodes;
    uint64_t local_in_edge_cnt, local_out_edge_cnt, global_edge_cnt;
    const uint32_t *local_to_global; /* N */
    const float *norms;              /* N   vtxDataVec */
    const uint32_t *src_ghosts;      /* Gsrc global ids, ascending (local id = N + k) */
    const uint32_t *dst_ghosts;      /* Gdst */
    const uint32_t *fwd_counts;      /* num_nodes: forwardGhostsList sizes */
    const uint32_t *fwd_lists;       /* concatenated local ids */
    const uint32_t *bwd_counts;
    const uint32_t *bwd_lists;
    const uint64_t *column_ptrs;     /* N+1  forwardAdj  */
    const uint32_t *row_idxs;
    const float *csc_values;
    const uint64_t *row_ptrs;        /* N+1  backwardAdj */
    const uint32_t *column_idxs;
    const float *csr_values;
};
int dory_partition_get(const dory_partition *p, struct dory_partition_view *view);

/* receive side of the halo plan for direction dir (0 forward / 1 backward): for each
 * peer q, recv_counts[q] rows arrive (in the sender's list order) and land in ghost
 * slots recv_slots[off_q .. off_q + recv_counts[q]) -- the ghost slots owned by q in
 * ascending global id.  recv_counts has num_nodes entries, recv_slots src/dst_ghost_cnt. */
int dory_partition_recv_plan(const dory_partition *p, const int32_t *parts, int dir,
                             uint32_t *recv_counts, uint32_t *recv_slots);

/* upload adjacency (dory_graph_upload) and, when parts is given, both halo plans
 * (dory_halo_plan): recv slots of peer q = ghost slots whose owner is q, ascending. */
int dory_partition_upload(dory_ctx *ctx, const dory_partition *p, const int32_t *parts);

/* ---- input files of the reference graph server (engine/utils.cpp:460-596) ------------- */
/* run/<dataset>.config: one layer width per line */
int dory_read_layer_config(const char *path, uint32_t *dims, uint32_t max_dims, uint32_t *count);
/* features.bsnap {u32 F} + V x F floats -> rows of this partition's local vertices
 * (N x F) and source ghosts (Gsrc x F); cache_dir (may be NULL) enables the reference's
 * feats<F>.<node>.bin cache in the dataset directory */
int dory_read_features(const char *path, const dory_partition *p, uint32_t expect_dim, uint32_t node_id,
                       const char *cache_dir, float *local, float *ghost);
/* labels.bsnap {u32 kinds} + V x u32 -> class id per local vertex */
int dory_read_labels(const char *path, const dory_partition *p, uint32_t expect_kinds, uint32_t *labels);
const char *dory_formats_last_error(void);

/* ---- synchronous-epoch Engine (the reference's stage order, one chunk per
 * partition: engine/engine.cpp:223-314, ops/pipeline.cpp, resource_comm.cpp) ------ */
typedef struct dory_engine dory_engine;
int dory_engine_create(dory_ctx *ctx, dory_engine **out);
int dory_engine_destroy(dory_engine *e);
/* run `epochs` epochs back to back; per-epoch wall time (ms, host clock around a
 * stream sync, as pipeline.cpp:118-127 measures consecutive epoch starts) is
 * written to epoch_ms[epochs] when non-NULL. */
int dory_engine_run(dory_engine *e, uint32_t epochs, double *epoch_ms);
/* single stages with the reference's Chunk semantics (for tests / foreign drivers) */
struct dory_chunk { /* common/utils.hpp:64-75 */
    uint32_t localId, globalId, lowBound, upBound, layer;
    int32_t dir;
    uint32_t epoch;
    uint8_t vertex;
};
int dory_engine_nn_compute(dory_engine *e, struct dory_chunk *chunk);   /* ResourceComm::NNCompute */
int dory_engine_inc_layer(dory_engine *e, const struct dory_chunk *in, struct dory_chunk *out); /* incLayerGCN/GAT */
int dory_engine_is_last_layer(dory_engine *e, const struct dory_chunk *c);
/* The same state machine without an engine or a device (engine/utils.cpp:707-753): next chunk of `in`. */
int dory_chunk_inc_layer(int gnn_type, uint32_t num_layers, const struct dory_chunk *in, struct dory_chunk *out);
/* Dry run of one synchronous epoch: the stages Engine::runEpoch would issue, as space-separated names
 * "<stage><layer><F|B>" with stage GA (aggregate), AV (applyVertex -> NNCompute), AE (applyEdge), SC (scatter),
 * plus "WU<layer>" (weight update sent, CPU_comm.cpp:131,147,178) and "PR<layer>" (predictGAT); no device needed.
 * The order is the one a chunk takes through the reference's queues (ops/pipeline.cpp:170-342,
 * resource_comm.cpp:17-90). */
int dory_engine_trace_epoch(int gnn_type, uint32_t num_layers, char *buf, size_t buflen);
/* "<EM>: ..." style report of the last run into buf (engine/utils.cpp:219-291) */
int dory_engine_report(dory_engine *e, char *buf, size_t buflen);

/* Layout introspection (tests): the destination side of the K1s layout -- dorylus_amd/host/sweep_deal.cpp.  `items` rows
 * (sorted by descending edge count) are dealt over 8 XCDs x S sweeps x sweep_tiles workgroups x 32 lane groups x
 * rows_per_group positions; groups of the last sweep carry fewer rows so that every sweep is whole.  positions_out: total
 * positions; group_rows_out (positions / rows_per_group entries, may be NULL): rows of every group; item_position
 * (`items` entries, may be NULL): the position of item i.  0 = ok. */
int dory_sweep_deal(uint32_t items, uint32_t rows_per_group, uint32_t sweep_tiles, uint32_t *positions_out,
                    uint32_t *group_rows_out, uint32_t *item_position);
/* The layout of the 32-lane launches with a loader wave (csrc/spmm.hip, LOADER): lane groups 0 and 1 of every workgroup --
 * the wave that also copies the next step's entries for the other fifteen -- get `loader_relief` rows fewer per sweep
 * (proportionally fewer in a shorter last sweep; never so many that a sweep more would be needed: then less or none), and
 * the items (sorted by descending weight = edge count, `weights` given) are dealt by weight, each to the group with the
 * smallest load per row slot, so that every group carries edges in proportion to its rows whatever the degree
 * distribution.  Outputs as dory_sweep_deal. */
int dory_sweep_deal_weighted(uint32_t items, const uint64_t *weights, uint32_t rows_per_group, uint32_t sweep_tiles,
                             uint32_t loader_relief, uint32_t *positions_out, uint32_t *group_rows_out,
                             uint32_t *item_position);

#ifdef __cplusplus
}
#endif
#endif /* DORYLUS_HOST_H */
