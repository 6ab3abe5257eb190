// hipcomm_adapter.cpp -- the reference-side glue of INTEGRATION.md sections 2 and 3, as one translation unit.
// What a Dorylus maintainer adds as commmanager/HIP_comm.{hpp,cpp} plus the `_HIP_ENABLED_` bodies in
// engine/ops/gcn_ops.cpp / gat_ops.cpp.  It is compiled here with -fsyntax-only against the reference's own headers
// (tests/test_hipcomm_adapter.py, only where /root/reference exists) to prove that the override matches
// ResourceComm (commmanager/resource_comm.hpp:13-28), that the Engine / Graph / Chunk / CommManager members it reads
// exist with the types the C-ABI takes, and that include/dorylus_hip.h + dorylus_wire.h parse as C++11 next to the
// reference's unscoped enums.  Nothing here is linked into the product.
//
// Multi-node wiring, all of it through members the reference already has:
//   * halo plan, send side  = Graph::forwardLocalVtxDsts / backwardLocalVtxDsts (graph/graph.cpp:52-66: the per-peer
//     lists of graph.<id>.bin);
//   * halo plan, receive side: every node tells each peer, once, the GLOBAL ids of the rows it will send it (in send
//     order) over the reference's control channel (CommManager::controlPushOut / controlPullIn, commmanager.hpp:40-41);
//     the peer turns them into ghost slots with Graph::srcGhostVtcs / dstGhostVtcs (graph/graph.cpp:34-49) --
//     what ghostReceiver does per message with the gvid prefix of every row (gcn_ops.cpp:284-318), done once;
//   * RCCL bootstrap: node 0 makes the 128-byte id, the control channel carries it;
//   * weights: replicated all-reduce + Adam on the GPUs (default), or -- HIPCommWS -- the unmodified weight servers:
//     pull with dory_wire_build_pull / dory_wire_parse_pull_reply -> dory_weight_set, push the gradient of
//     dory_weight_grad_get with dory_wire_build_push (commmanager/message_service.cpp:40-108 frame for frame).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "commmanager/resource_comm.hpp"
#include "engine/engine.hpp"

#include "dorylus_hip.h"
#include "dorylus_wire.h"

static inline int hipDir(const Chunk &c) { return c.dir == PROP_TYPE::FORWARD ? DORY_FORWARD : DORY_BACKWARD; }

class HIPComm : public ResourceComm {
public:
    explicit HIPComm(Engine *e) : ctx(NULL), engine(e) {
        dory_create(/*device*/ (int)(e->nodeId % 8), &ctx);
        std::vector<uint32_t> dims(e->layerConfig.begin(), e->layerConfig.end());
        dory_configure(ctx, e->gnn_type == GNN::GCN ? DORY_GCN : DORY_GAT, e->numLayers, dims.data(),
                       e->graph.globalVtxCnt, e->nodeId, e->numNodes);
        Graph &g = e->graph;   // the arrays Graph::init read from graph.<id>.bin (graph/graph.cpp:7-115)
        static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "columnPtrs / rowPtrs are 64-bit");
        dory_graph_upload(ctx, g.localVtxCnt, g.srcGhostCnt, g.dstGhostCnt,
                          g.forwardAdj.nnz, reinterpret_cast<const uint64_t *>(g.forwardAdj.columnPtrs),
                          g.forwardAdj.rowIdxs, g.forwardAdj.values,
                          g.backwardAdj.nnz, reinterpret_cast<const uint64_t *>(g.backwardAdj.rowPtrs),
                          g.backwardAdj.columnIdxs, g.backwardAdj.values, g.vtxDataVec.data());
        dory_preallocate(ctx);   // the names of preallocateGCN / preallocateGAT (gcn_ops.cpp:27-93, gat_ops.cpp:27-115)
        dory_tensor_upload(ctx, 0, e->gnn_type == GNN::GCN ? "x" : "h", e->forwardVerticesInitData);
        if (e->gnn_type == GNN::GCN && g.srcGhostCnt) dory_tensor_upload(ctx, 0, "fg", e->forwardGhostInitData);
        dory_tensor_upload(ctx, e->numLayers - 1, "lab", e->localVerticesLabels);
        if (e->numNodes > 1) {
            planHalo(DORY_FORWARD, g.forwardLocalVtxDsts, g.srcGhostVtcs);
            planHalo(DORY_BACKWARD, g.backwardLocalVtxDsts, g.dstGhostVtcs);
            initComm();
        }
    }
    ~HIPComm() { dory_destroy(ctx); }

    // CPUComm::NNCompute / GPUComm::NNCompute contract (CPU_comm.cpp:22-44, GPU_comm.cpp:11-35)
    void NNCompute(Chunk &c) override {
        if (c.vertex) dory_apply_vertex(ctx, c.layer, hipDir(c));
        else          dory_apply_edge(ctx, c.layer, hipDir(c));
        // the gradient of this stage leaves for the "weight server" right after the stage that produced it
        // (CPU_comm.cpp:131,147,178): here = RCCL all-reduce over the node's GPUs + Adam on every replica
        const bool lastFwd = c.vertex && c.dir == PROP_TYPE::FORWARD && c.layer == engine->numLayers - 1 &&
                             engine->gnn_type == GNN::GCN;
        if (c.vertex && (c.dir == PROP_TYPE::BACKWARD || lastFwd)) pushGradient(c);
        NNRecvCallback(engine, c);
    }
    unsigned getRelaunchCnt() override { return 0u; }

    dory_ctx *ctx;
    Engine *engine;

protected:
    virtual void pushGradient(Chunk &c) { dory_weight_update(ctx, c.layer); }

private:
    // One direction of the halo plan.  sendLists[p] = my local rows that peer p reads (file order = ascending local
    // id); ghostSlots = gvid -> local id (N + k) of the rows I read from others.
    void planHalo(int dir, const std::vector<std::vector<unsigned>> &sendLists, const std::map<unsigned, unsigned> &ghostSlots) {
        Engine *e = engine;
        Graph &g = e->graph;
        const unsigned P = e->numNodes, me = e->nodeId;
        std::vector<uint32_t> sendCnt(P, 0), sendLvids, recvCnt(P, 0), recvSlots;
        for (unsigned p = 0; p < P; ++p) {
            const std::vector<unsigned> &l = sendLists[p];
            sendCnt[p] = p == me ? 0u : (uint32_t)l.size();
            if (p == me) continue;
            sendLvids.insert(sendLvids.end(), l.begin(), l.end());
            // announce, once, the global ids in send order: {count, gvid...}
            std::vector<unsigned> msg(1 + l.size());
            msg[0] = (unsigned)l.size();
            for (size_t i = 0; i < l.size(); ++i) msg[1 + i] = g.localToGlobalId[l[i]];
            e->commManager.controlPushOut(p, msg.data(), (unsigned)(msg.size() * sizeof(unsigned)));
        }
        std::vector<unsigned> in(1 + (dir == DORY_FORWARD ? g.srcGhostCnt : g.dstGhostCnt));
        for (unsigned p = 0; p < P; ++p) {
            if (p == me) continue;
            while (!e->commManager.controlPullIn(p, in.data(), (unsigned)(in.size() * sizeof(unsigned)))) { /* spin like Engine::init's barrier */ }
            recvCnt[p] = in[0];
            for (unsigned i = 0; i < in[0]; ++i) {
                std::map<unsigned, unsigned>::const_iterator it = ghostSlots.find(in[1 + i]);
                recvSlots.push_back(it == ghostSlots.end() ? 0u : it->second - g.localVtxCnt);   // ghost local id N + k -> slot k
            }
        }
        dory_halo_plan(ctx, dir, sendCnt.data(), sendLvids.data(), recvCnt.data(), recvSlots.data());
    }

    void initComm() {
        Engine *e = engine;
        unsigned char id[128];
        if (e->nodeId == 0) {
            dory_comm_unique_id(id);
            for (unsigned p = 1; p < e->numNodes; ++p) e->commManager.controlPushOut(p, id, sizeof(id));
        } else {
            while (!e->commManager.controlPullIn(0, id, sizeof(id))) { }
        }
        dory_comm_init(ctx, id, (int)e->nodeId, (int)e->numNodes);
    }
};

// ---- variant: keep the reference's weight servers (a mixed deployment) ------------------------------------------------
// Weights are pulled before a vertex stage and the stage's gradient is pushed after it, with the byte formats of
// message_service.cpp:40-108 built by dorylus_wire.h; `wsocket` is a DEALER socket connected to a weight server the way
// MessageService connects its own (message_service.cpp:115-140).
class HIPCommWS : public HIPComm {
public:
    HIPCommWS(Engine *e, zmq::socket_t *ws) : HIPComm(e), wsocket(ws) {}

    void NNCompute(Chunk &c) override {
        if (c.vertex) pullWeights(c);
        HIPComm::NNCompute(c);
    }

protected:
    void pushGradient(Chunk &c) override {
        const uint32_t rows = engine->layerConfig[c.layer], cols = engine->layerConfig[c.layer + 1];
        std::vector<float> grad((size_t)rows * cols);
        dory_weight_grad_get(ctx, c.layer, "w", grad.data());
        dory_wire_chunk wc = wireChunk(c);
        const char *names[1] = {"w"};
        const float *data[1] = {grad.data()};
        std::vector<uint8_t> buf(DORY_WIRE_HEADER_SIZE + DORY_WIRE_TENSOR_HDR_SIZE + grad.size() * sizeof(float));
        size_t off[4];
        const int nf = dory_wire_build_push(&wc, names, &rows, &cols, data, 1, buf.data(), buf.size(), off, 3);
        for (int f = 0; f < nf; ++f) {   // sendTensors (message_service.cpp:79-108): every frame but the last with SNDMORE
            zmq::message_t m(off[f + 1] - off[f]);
            std::memcpy(m.data(), buf.data() + off[f], off[f + 1] - off[f]);
            wsocket->send(m, f + 1 < nf ? ZMQ_SNDMORE : 0);
        }
    }

private:
    static dory_wire_chunk wireChunk(const Chunk &c) {
        dory_wire_chunk w;
        w.local_id = c.localId; w.global_id = c.globalId; w.low_bound = c.lowBound; w.up_bound = c.upBound;
        w.layer = c.layer; w.dir = c.dir == PROP_TYPE::FORWARD ? 0 : 1; w.epoch = c.epoch; w.vertex = c.vertex ? 1 : 0;
        return w;
    }

    void pullWeights(Chunk &c) {   // reqTensors + recvTensor (message_service.cpp:17-77)
        dory_wire_chunk wc = wireChunk(c);
        const char *names[1] = {"w"};
        uint8_t buf[DORY_WIRE_HEADER_SIZE + DORY_WIRE_TENSOR_HDR_SIZE];
        size_t off[3];
        const int nf = dory_wire_build_pull(&wc, names, 1, buf, sizeof(buf), off, 2);
        for (int f = 0; f < nf; ++f) {
            zmq::message_t m(off[f + 1] - off[f]);
            std::memcpy(m.data(), buf + off[f], off[f + 1] - off[f]);
            wsocket->send(m, f + 1 < nf ? ZMQ_SNDMORE : 0);
        }
        zmq::message_t hdr, payload;
        wsocket->recv(&hdr);
        char name[9];
        uint32_t rows = 0, cols = 0;
        if (hdr.size() != DORY_WIRE_TENSOR_HDR_SIZE) return;
        unsigned more = 0;
        size_t usize = sizeof(more);
        wsocket->getsockopt(ZMQ_RCVMORE, &more, &usize);
        if (more) wsocket->recv(&payload);
        if (dory_wire_parse_pull_reply(hdr.data(), payload.size(), name, &rows, &cols) == 0)
            dory_weight_set(ctx, c.layer, name, static_cast<const float *>(payload.data()));
    }

    zmq::socket_t *wsocket;
};

static HIPComm *hipComm(Engine *e) { return static_cast<HIPComm *>(e->resComm); }

// the `_HIP_ENABLED_` bodies next to the `_GPU_ENABLED_` ones (gcn_ops.cpp:95-128, gat_ops.cpp:118-171)
void Engine::aggregateGCN(Chunk &c) { dory_aggregate(hipComm(this)->ctx, c.layer, hipDir(c)); }
void Engine::aggregateGAT(Chunk &c) { dory_aggregate(hipComm(this)->ctx, c.layer, hipDir(c)); }
void Engine::scatterGCN(Chunk &c) { dory_halo_exchange(hipComm(this)->ctx, c.layer, hipDir(c)); }
void Engine::scatterGAT(Chunk &c) { dory_halo_exchange(hipComm(this)->ctx, c.layer, hipDir(c)); }

// engine/engine.cpp:141-163 gains:  #elif defined(_HIP_ENABLED_)   resComm = new HIPComm(this);
ResourceComm *make_hip_comm(Engine *e) { return new HIPComm(e); }
ResourceComm *make_hip_comm_ws(Engine *e, zmq::socket_t *weightServer) { return new HIPCommWS(e, weightServer); }
