// hipcomm_adapter.cpp -- the reference-side glue of INTEGRATION.md sections 2 and 3, as one translation unit.
// What a Dorylus maintainer adds as commmanager/HIP_comm.{hpp,cpp} plus the `_HIP_ENABLED_` bodies in
// engine/ops/gcn_ops.cpp / gat_ops.cpp.  It is compiled here with -fsyntax-only against the reference's own headers
// (tests/test_hipcomm_adapter.py, only where /root/reference exists) to prove that the override matches
// ResourceComm (commmanager/resource_comm.hpp:13-28), that the Engine / Graph / Chunk members it reads exist with the
// types the C-ABI takes, and that include/dorylus_hip.h parses as C++11 next to the reference's unscoped enums.
// Nothing here is linked into the product.
#include <cstdint>
#include <vector>

#include "commmanager/resource_comm.hpp"
#include "engine/engine.hpp"

#include "dorylus_hip.h"

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
        // halo plan: the per-peer send lists of the partition file and the ghost slots each peer's rows land in
        std::vector<uint32_t> cnt(e->numNodes), lvids;
        for (unsigned p = 0; p < e->numNodes; ++p) {
            const std::vector<unsigned> &l = g.forwardLocalVtxDsts[p];
            cnt[p] = (uint32_t)l.size();
            lvids.insert(lvids.end(), l.begin(), l.end());
        }
        (void)cnt; (void)lvids;   // + recv_counts / recv_slots from srcGhostVtcs and the .parts vector: dory_partition_upload does all of it
    }
    ~HIPComm() { dory_destroy(ctx); }

    // CPUComm::NNCompute / GPUComm::NNCompute contract (CPU_comm.cpp:22-44, GPU_comm.cpp:11-35)
    void NNCompute(Chunk &c) override {
        if (c.vertex) dory_apply_vertex(ctx, c.layer, c.dir == PROP_TYPE::FORWARD ? DORY_FORWARD : DORY_BACKWARD);
        else          dory_apply_edge(ctx, c.layer, c.dir == PROP_TYPE::FORWARD ? DORY_FORWARD : DORY_BACKWARD);
        if (c.vertex && c.dir == PROP_TYPE::BACKWARD) dory_weight_update(ctx, c.layer);   // or: dory_weight_grad_get -> the weight server
        NNRecvCallback(engine, c);
    }
    unsigned getRelaunchCnt() override { return 0u; }

    dory_ctx *ctx;
    Engine *engine;
};

static HIPComm *hipComm(Engine *e) { return static_cast<HIPComm *>(e->resComm); }

// the `_HIP_ENABLED_` bodies next to the `_GPU_ENABLED_` ones (gcn_ops.cpp:95-128, gat_ops.cpp:118-171)
void Engine::aggregateGCN(Chunk &c) { dory_aggregate(hipComm(this)->ctx, c.layer, c.dir == PROP_TYPE::FORWARD ? DORY_FORWARD : DORY_BACKWARD); }
void Engine::aggregateGAT(Chunk &c) { dory_aggregate(hipComm(this)->ctx, c.layer, c.dir == PROP_TYPE::FORWARD ? DORY_FORWARD : DORY_BACKWARD); }
void Engine::scatterGCN(Chunk &c) { dory_halo_exchange(hipComm(this)->ctx, c.layer, c.dir == PROP_TYPE::FORWARD ? DORY_FORWARD : DORY_BACKWARD); }
void Engine::scatterGAT(Chunk &c) { dory_halo_exchange(hipComm(this)->ctx, c.layer, c.dir == PROP_TYPE::FORWARD ? DORY_FORWARD : DORY_BACKWARD); }

// engine/engine.cpp:141-163 gains:  #elif defined(_HIP_ENABLED_)   resComm = new HIPComm(this);
ResourceComm *make_hip_comm(Engine *e) { return new HIPComm(e); }
