// engine.cpp -- host mirror of the reference Engine's synchronous epoch for the
// hot path, in C++ like the reference (src/graph-server/engine).  Same stage
// order (scheduler -> GA -> AV -> SC -> AE), same Chunk state machine
// (incLayerGCN/GAT, isLastLayer: engine/utils.cpp:707-753), same callback contract
// (ResourceComm::NNCompute must advance the chunk: commmanager/resource_comm.cpp:4-90).
// In cpu/gpu mode numlambdas is forced to 1 (run/run-onnode:62-70), i.e. one Chunk
// [0, N) per partition; the thread pipeline and staleness machinery of the Lambda
// mode are out of scope.  Every stage is one C-ABI call.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dorylus_host.h"

namespace dorylus {

struct Chunk {  // common/utils.hpp:64-75
    unsigned localId, globalId, lowBound, upBound, layer;
    int dir;
    unsigned epoch;
    bool vertex;
};

class Engine;

// The reference's abstract backend (commmanager/resource_comm.hpp:13-28).
class ResourceComm {
  public:
    virtual ~ResourceComm() {}
    virtual int NNCompute(Chunk &chunk) = 0;
    virtual void prefetchWeights() {}
    virtual unsigned getRelaunchCnt() { return 0u; }
};

// HIPComm: the MI355X backend behind the ResourceComm boundary -- what
// CPUComm::NNCompute (commmanager/CPU_comm.cpp:22-44) / GPUComm::NNCompute
// (GPU_comm.cpp:11-35) are for the cpu / gpu builds.
class HIPComm : public ResourceComm {
  public:
    HIPComm(dory_ctx *ctx, int gnn, unsigned totalLayers, std::vector<std::string> *trace = nullptr)
        : ctx_(ctx), gnn_(gnn), totalLayers_(totalLayers), trace_(trace) {}
    int NNCompute(Chunk &chunk) override {
        int rc;
        if (trace_) {   // dry run (dory_engine_trace_epoch): the same decisions, names instead of launches
            const bool produced = chunk.vertex && (gnn_ == DORY_GCN ? (chunk.dir == DORY_BACKWARD || chunk.layer == totalLayers_ - 1)
                                                                    : (chunk.dir == DORY_BACKWARD));
            trace_->push_back(stage_name(chunk.vertex ? "AV" : "AE", chunk.layer, chunk.dir));
            if (produced) trace_->push_back("WU" + std::to_string(chunk.layer));
            return DORY_OK;
        }
        if (chunk.vertex) {
            if ((rc = dory_apply_vertex(ctx_, chunk.layer, chunk.dir))) return rc;
            // sendWeightUpdate (CPU_comm.cpp:131,147,178): the gradient leaves for the
            // weight server as soon as the stage has produced it; here that is the
            // RCCL all-reduce + Adam step on the device.
            bool produced = gnn_ == DORY_GCN
                                ? (chunk.dir == DORY_BACKWARD || chunk.layer == totalLayers_ - 1)
                                : (chunk.dir == DORY_BACKWARD);
            // transform-first order of this layer: dW_l only exists after the backward aggregation of layer l
            // (Engine::runEpoch sends the update then)
            if (produced && dory_transform_first_layer(ctx_, chunk.layer)) produced = false;
            if (produced && (rc = dory_weight_update(ctx_, chunk.layer))) return rc;
            return DORY_OK;
        }
        return dory_apply_edge(ctx_, chunk.layer, chunk.dir);  // layer-- happens inside (CPU_comm.cpp:33)
    }

    static std::string stage_name(const char *stage, unsigned layer, int dir) {
        return std::string(stage) + std::to_string(layer) + (dir == DORY_FORWARD ? "F" : "B");
    }

  private:
    dory_ctx *ctx_;
    int gnn_;
    unsigned totalLayers_;
    std::vector<std::string> *trace_;
};

class Engine {
  public:
    Engine(dory_ctx *ctx, int gnn, unsigned numLayers, unsigned nodeId, unsigned localVtxCnt,
           std::vector<std::string> *trace = nullptr)
        : ctx(ctx), gnn_type(gnn), numLayers(numLayers), nodeId(nodeId), localVtxCnt(localVtxCnt),
          resComm(new HIPComm(ctx, gnn, numLayers, trace)), trace(trace) {}
    ~Engine() { delete resComm; }

    // ---- layer utils (engine/utils.cpp:707-753) ------------------------------------
    Chunk incLayerGCN(const Chunk &chunk) const {
        Chunk n = chunk;
        if (n.dir == DORY_FORWARD) {
            n.layer++;
            if (n.layer == numLayers) {  // merge forward and backward pass of the last layer
                n.dir = DORY_BACKWARD;
                n.layer--;
            }
        } else {
            if (n.layer == 0) {
                n.dir = DORY_FORWARD;
                n.epoch++;
            } else {
                n.layer--;
            }
        }
        return n;
    }
    Chunk incLayerGAT(const Chunk &chunk) const {
        Chunk n = chunk;
        if (n.dir == DORY_FORWARD) {
            n.layer++;
        } else {
            if (n.layer == 0) {
                n.dir = DORY_FORWARD;
                n.vertex = true;
                n.epoch++;
            } else {
                n.layer--;
            }
        }
        return n;
    }
    Chunk incLayer(const Chunk &c) const { return gnn_type == DORY_GCN ? incLayerGCN(c) : incLayerGAT(c); }
    bool isLastLayer(const Chunk &c) const { return c.dir == DORY_BACKWARD && c.layer == 0 && c.vertex; }

    // ---- SAGA stages -----------------------------------------------------------------
    int stage(const char *name, const Chunk &c, int (*call)(dory_ctx *, uint32_t, int)) {
        if (trace) {
            trace->push_back(HIPComm::stage_name(name, c.layer, c.dir));
            return DORY_OK;
        }
        return call(ctx, c.layer, c.dir);
    }
    int aggregateGCN(Chunk &c) { return stage("GA", c, dory_aggregate); }   // gcn_ops.cpp:130-191
    int aggregateGAT(Chunk &c) { return stage("GA", c, dory_aggregate); }   // gat_ops.cpp:173-243
    int applyVertexGCN(Chunk &c) {                                               // gcn_ops.cpp:194-202
        c.vertex = true;
        if (c.dir == DORY_FORWARD) return resComm->NNCompute(c);
        Chunk nextC = incLayerGCN(c);
        int rc = resComm->NNCompute(nextC);
        c = nextC;
        return rc;
    }
    int applyVertexGAT(Chunk &c) {                                               // gat_ops.cpp:267-275
        c.vertex = true;
        if (c.dir == DORY_FORWARD) return resComm->NNCompute(c);
        Chunk nextC = incLayerGAT(c);
        int rc = resComm->NNCompute(nextC);
        c = nextC;
        return rc;
    }
    int scatterGCN(Chunk &c) { return stage("SC", c, dory_halo_exchange); }  // gcn_ops.cpp:204-362
    int scatterGAT(Chunk &c) { return stage("SC", c, dory_halo_exchange); }  // gat_ops.cpp:277-435
    int applyEdgeGAT(Chunk &c) {                                                  // gat_ops.cpp:437-440
        c.vertex = false;
        return resComm->NNCompute(c);
    }
    int predictGAT(Chunk &c) {                                                    // gat_ops.cpp:246-265
        if (trace) {
            trace->push_back("PR" + std::to_string(c.layer));
            return DORY_OK;
        }
        return dory_predict_gat(ctx, c.layer);
    }

    // One synchronous epoch: the path a chunk takes through the queues
    // (ops/pipeline.cpp:170-173,183-219,262-342; resource_comm.cpp:17-51,53-90).
    int runEpoch(unsigned epoch) {
        Chunk c{0, nodeId, 0, localVtxCnt, 0, DORY_FORWARD, epoch, true};
        int rc;
        // a one-layer GCN never reaches isLastLayer in this state machine (its only forward layer
        // merges into a backward pass that re-enters the forward one, engine/utils.cpp:707-727):
        // the reference would spin through epochs without a boundary; refuse instead
        if (gnn_type == DORY_GCN && numLayers < 2) return DORY_ERR_ARG;
        if (gnn_type == DORY_GCN) {
            for (;;) {
                if ((rc = aggregateGCN(c))) return rc;          // GA
                if (!trace && c.dir == DORY_BACKWARD && dory_transform_first_layer(ctx, c.layer) &&
                    (rc = dory_weight_update(ctx, c.layer)))    // transform-first: this aggregation produced dW_l
                    return rc;
                if ((rc = applyVertexGCN(c))) return rc;        // AV -> NNRecvCallbackGCN
                if (isLastLayer(c)) {                           //   -> schQueue (next epoch)
                    if (!trace && dory_transform_first_layer(ctx, 0)) {
                        // transform-first order of layer 0 (dorylus_hip.h): dW0 = X^T (A^T g0) needs g0's ghost
                        // rows and one more aggregation on the out-edges before the update can leave
                        if ((rc = scatterGCN(c))) return rc;
                        if ((rc = aggregateGCN(c))) return rc;
                        if ((rc = dory_weight_update(ctx, 0))) return rc;
                    }
                    break;
                }
                if (c.dir == DORY_FORWARD) c = incLayerGCN(c);  //   forward: inc layer after AV
                if ((rc = scatterGCN(c))) return rc;            // SC (+ ghostReceiver, barrier)
                // AE: applyEdgeGCN forwards to GA (gcn_ops.cpp:364-366)
            }
        } else {
            for (;;) {
                if ((rc = applyVertexGAT(c))) return rc;        // AV
                if (isLastLayer(c)) break;
                if (c.dir == DORY_FORWARD) c = incLayerGAT(c);
                if ((rc = scatterGAT(c))) return rc;            // SC
                if ((rc = applyEdgeGAT(c))) return rc;          // AE
                if ((rc = aggregateGAT(c))) return rc;          // GA
                if (c.dir == DORY_FORWARD && c.layer == numLayers) {  // pipeline.cpp:203-213
                    if ((rc = predictGAT(c))) return rc;
                    c.dir = DORY_BACKWARD;
                    if ((rc = scatterGAT(c))) return rc;
                    if ((rc = applyEdgeGAT(c))) return rc;
                    if ((rc = aggregateGAT(c))) return rc;
                }
            }
        }
        return DORY_OK;
    }

    dory_ctx *ctx;
    int gnn_type;
    unsigned numLayers, nodeId, localVtxCnt;
    ResourceComm *resComm;
    std::vector<std::string> *trace;   // dry run: stage names instead of C-ABI calls
    std::vector<double> epochTimes;
};

}  // namespace dorylus

struct dory_engine {
    dorylus::Engine *eng;
    unsigned nextEpoch = 1;  // START_EPOCH + 1 (engine/utils.cpp:607)
    unsigned numNodes = 1;
    bool warmed = false;     // one eager epoch has run (lazy allocations exist)
    bool recorded = false;   // the ctx holds a recorded epoch
};

// the engine needs a few facts the ctx already knows; they are exported by csrc/abi_context.hip
extern "C" int dory_ctx_describe(dory_ctx *ctx, int *gnn, uint32_t *num_layers, uint32_t *node_id,
                                 uint32_t *num_nodes, uint32_t *local_vtx_cnt);

extern "C" {

int dory_engine_create(dory_ctx *ctx, dory_engine **out) {
    if (!ctx || !out) return DORY_ERR_ARG;
    int gnn;
    uint32_t L, nodeId, numNodes, N;
    int rc = dory_ctx_describe(ctx, &gnn, &L, &nodeId, &numNodes, &N);
    if (rc) return rc;
    dory_engine *e = new dory_engine();
    e->eng = new dorylus::Engine(ctx, gnn, L, nodeId, N);
    e->numNodes = numNodes;
    *out = e;
    return DORY_OK;
}

int dory_engine_destroy(dory_engine *e) {
    if (!e) return DORY_ERR_ARG;
    delete e->eng;
    delete e;
    return DORY_OK;
}

int dory_engine_run(dory_engine *e, uint32_t epochs, double *epoch_ms) {
    if (!e) return DORY_ERR_ARG;
    int rc = dory_sync(e->eng->ctx);
    if (rc) return rc;
    int64_t want_graph = 0;
    if (dory_get_option(e->eng->ctx, "epoch_graph", &want_graph)) want_graph = 0;
    if (e->numNodes > 1) want_graph = 0;   // the exchange is not recorded
    if (e->recorded) {   // dory_preallocate / dory_graph_upload drop a recorded epoch behind the engine's back
        int64_t have = 0;
        if (dory_get_option(e->eng->ctx, "epoch_graph_recorded", &have) || !have) {
            e->recorded = false;
            e->warmed = false;   // a new graph / tensor table: lazily sized buffers need one eager epoch again before recording
        }
    }
    if (!want_graph && e->recorded) {
        dory_epoch_graph_drop(e->eng->ctx);
        e->recorded = false;
    }
    for (uint32_t i = 0; i < epochs; ++i) {
        auto t0 = std::chrono::steady_clock::now();
        if (want_graph && e->warmed) {
            // epoch graph: record the epoch once (the same calls, captured instead of run), then replay
            if (!e->recorded) {
                if ((rc = dory_epoch_graph_begin(e->eng->ctx))) return rc;
                rc = e->eng->runEpoch(e->nextEpoch);
                if (rc) {
                    dory_epoch_graph_drop(e->eng->ctx);
                    return rc;
                }
                if ((rc = dory_epoch_graph_end(e->eng->ctx))) return rc;
                e->recorded = true;
                t0 = std::chrono::steady_clock::now();
            }
            if ((rc = dory_epoch_graph_launch(e->eng->ctx, 1))) return rc;
        } else if ((rc = e->eng->runEpoch(e->nextEpoch))) {
            return rc;
        }
        e->warmed = true;
        // epoch boundary = the scheduler's barrier (pipeline.cpp:103-127): all local
        // work of the epoch has finished
        if ((rc = dory_sync(e->eng->ctx))) return rc;
        auto t1 = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        e->eng->epochTimes.push_back(ms);
        if (epoch_ms) epoch_ms[i] = ms;
        ++e->nextEpoch;
    }
    return DORY_OK;
}

int dory_engine_nn_compute(dory_engine *e, struct dory_chunk *ch) {
    if (!e || !ch) return DORY_ERR_ARG;
    dorylus::Chunk c{ch->localId, ch->globalId, ch->lowBound, ch->upBound, ch->layer, ch->dir, ch->epoch, ch->vertex != 0};
    return e->eng->resComm->NNCompute(c);
}

int dory_engine_inc_layer(dory_engine *e, const struct dory_chunk *in, struct dory_chunk *out) {
    if (!e || !in || !out) return DORY_ERR_ARG;
    dorylus::Chunk c{in->localId, in->globalId, in->lowBound, in->upBound, in->layer, in->dir, in->epoch, in->vertex != 0};
    dorylus::Chunk n = e->eng->incLayer(c);
    *out = dory_chunk{n.localId, n.globalId, n.lowBound, n.upBound, n.layer, n.dir, n.epoch, (uint8_t)n.vertex};
    return DORY_OK;
}

int dory_engine_is_last_layer(dory_engine *e, const struct dory_chunk *in) {
    if (!e || !in) return DORY_ERR_ARG;
    dorylus::Chunk c{in->localId, in->globalId, in->lowBound, in->upBound, in->layer, in->dir, in->epoch, in->vertex != 0};
    return e->eng->isLastLayer(c) ? 1 : 0;
}

int dory_engine_trace_epoch(int gnn_type, uint32_t num_layers, char *buf, size_t buflen) {
    if ((gnn_type != DORY_GCN && gnn_type != DORY_GAT) || num_layers == 0 || !buf || buflen == 0) return DORY_ERR_ARG;
    std::vector<std::string> tr;
    dorylus::Engine eng(nullptr, gnn_type, num_layers, 0, 0, &tr);
    int rc = eng.runEpoch(1);
    if (rc) return rc;
    std::string out;
    for (size_t i = 0; i < tr.size(); ++i) out += (i ? " " : "") + tr[i];
    if (out.size() + 1 > buflen) return DORY_ERR_ARG;
    memcpy(buf, out.c_str(), out.size() + 1);
    return DORY_OK;
}

int dory_chunk_inc_layer(int gnn_type, uint32_t num_layers, const struct dory_chunk *in, struct dory_chunk *out) {
    if ((gnn_type != DORY_GCN && gnn_type != DORY_GAT) || num_layers == 0 || !in || !out) return DORY_ERR_ARG;
    dorylus::Engine eng(nullptr, gnn_type, num_layers, 0, 0);
    dorylus::Chunk c{in->localId, in->globalId, in->lowBound, in->upBound, in->layer, in->dir, in->epoch, in->vertex != 0};
    dorylus::Chunk n = eng.incLayer(c);
    *out = dory_chunk{n.localId, n.globalId, n.lowBound, n.upBound, n.layer, n.dir, n.epoch, (uint8_t)n.vertex};
    return DORY_OK;
}

int dory_engine_report(dory_engine *e, char *buf, size_t buflen) {
    if (!e || !buf || buflen == 0) return DORY_ERR_ARG;
    // "<EM>: Average  sync epoch time %.3lf ms" (engine/utils.cpp:283-288); epoch 0 is
    // skipped by the reference's epoch timer (pipeline.cpp:118-127)
    const auto &t = e->eng->epochTimes;
    double sum = 0;
    size_t n = 0;
    for (size_t i = 1; i < t.size(); ++i) { sum += t[i]; ++n; }
    snprintf(buf, buflen,
             "[ Node %3u ]  <EM>: Run start time: n/a\n"
             "[ Node %3u ]  <EM>: Using %u forward lambdas and %u bacward lambdas\n"
             "[ Node %3u ]  <EM>: Backend HIP\n"
             "[ Node %3u ]  <EM>: %zu sync epochs and 0 async epochs\n"
             "[ Node %3u ]  <EM>: Average  sync epoch time %.3lf ms\n",
             e->eng->nodeId, e->eng->nodeId, 1u, 1u, e->eng->nodeId, e->eng->nodeId, t.size(), e->eng->nodeId,
             n ? sum / n : 0.0);
    return DORY_OK;
}

}  // extern "C"
