"""Host-side mirror of the reference Engine's synchronous epoch for the hot path.

Same stage order, chunk state machine and entry-point names as the reference
(src/graph-server/engine): scheduler -> GA (aggregate*) -> AV (applyVertex* ->
ResourceComm::NNCompute) -> SC (scatter*) -> AE (applyEdge*) -> GA ...; in cpu/gpu
mode there is exactly one chunk [0, N) per partition (engine/utils.cpp:598-609).
Every stage is one C-ABI call; nothing here computes.
"""
import ctypes as C
from dataclasses import dataclass, replace

import numpy as np

from ._lib import BACKWARD, FORWARD, GAT, GCN, Context, DoryError, load


@dataclass
class Chunk:
    """common/utils.hpp:64-75"""
    localId: int = 0
    globalId: int = 0
    lowBound: int = 0
    upBound: int = 0
    layer: int = 0
    dir: int = FORWARD
    epoch: int = 0
    vertex: bool = True


class _CChunk(C.Structure):
    """struct dory_chunk (include/dorylus_host.h)"""
    _fields_ = [("localId", C.c_uint32), ("globalId", C.c_uint32), ("lowBound", C.c_uint32), ("upBound", C.c_uint32),
                ("layer", C.c_uint32), ("dir", C.c_int32), ("epoch", C.c_uint32), ("vertex", C.c_uint8)]


class Engine:
    def __init__(self, ctx: Context, gnn: int, num_layers: int):
        self.ctx = ctx
        self.gnn_type = gnn
        self.numLayers = num_layers
        self.epoch_hook = None

    # ---- layer utils (engine/utils.cpp:707-753): ONE implementation, host/engine.cpp's, reached through the C-ABI
    # (dory_chunk_inc_layer needs neither an engine nor a device) -- this class keeps no copy of the state machine
    def _inc_layer(self, gnn, c: Chunk) -> Chunk:
        cin = _CChunk(c.localId, c.globalId, c.lowBound, c.upBound, c.layer, c.dir, c.epoch, 1 if c.vertex else 0)
        cout = _CChunk()
        lib = getattr(self.ctx, "lib", None) or load()      # (a recording stand-in for the context in tests has no library handle)
        rc = lib.dory_chunk_inc_layer(gnn, self.numLayers, C.byref(cin), C.byref(cout))
        if rc != 0:
            raise DoryError(f"dory_chunk_inc_layer failed ({rc})")
        return Chunk(cout.localId, cout.globalId, cout.lowBound, cout.upBound, cout.layer, cout.dir, cout.epoch, bool(cout.vertex))

    def incLayerGCN(self, c: Chunk) -> Chunk:
        return self._inc_layer(GCN, c)

    def incLayerGAT(self, c: Chunk) -> Chunk:
        return self._inc_layer(GAT, c)

    def isLastLayer(self, c: Chunk) -> bool:
        return c.dir == BACKWARD and c.layer == 0 and c.vertex

    # ---- SAGA stages --------------------------------------------------------------------
    def aggregateGCN(self, c):     # engine/ops/gcn_ops.cpp:130-191
        self.ctx.aggregate(c.layer, c.dir)

    def aggregateGAT(self, c):     # engine/ops/gat_ops.cpp:173-243
        self.ctx.aggregate(c.layer, c.dir)

    def NNCompute(self, c):        # ResourceComm::NNCompute (commmanager/CPU_comm.cpp:22-44)
        if c.vertex:
            self.ctx.apply_vertex(c.layer, c.dir)
            # weight updates leave for the "weight server" right after the stage that
            # produced them (CPU_comm.cpp:131,147): all-reduce + Adam on the device
            if self.gnn_type == GCN:
                if (c.dir == BACKWARD or c.layer == self.numLayers - 1) and not self._transform_first(c.layer):
                    self.ctx.weight_update(c.layer)          # transform-first layers: after their backward aggregation
            elif c.dir == BACKWARD:
                self.ctx.weight_update(c.layer)
        else:
            self.ctx.apply_edge(c.layer, c.dir)

    def _transform_first(self, layer):
        f = getattr(self.ctx, "transform_first_layer", None)
        return bool(f(layer)) if f else False

    def applyVertexGCN(self, c):   # gcn_ops.cpp:194-202
        c.vertex = True
        if c.dir == FORWARD:
            self.NNCompute(c)
            return c
        nxt = self.incLayerGCN(c)
        self.NNCompute(nxt)
        return nxt

    def applyVertexGAT(self, c):   # gat_ops.cpp:267-275
        c.vertex = True
        if c.dir == FORWARD:
            self.NNCompute(c)
            return c
        nxt = self.incLayerGAT(c)
        self.NNCompute(nxt)
        return nxt

    def scatterGCN(self, c):       # gcn_ops.cpp:204-260 + ghostReceiverGCN :262-362
        self.ctx.halo_exchange(c.layer, c.dir)

    def scatterGAT(self, c):       # gat_ops.cpp:277-340
        self.ctx.halo_exchange(c.layer, c.dir)

    def applyEdgeGAT(self, c):     # gat_ops.cpp:437-440
        c.vertex = False
        self.NNCompute(c)

    # ---- one synchronous epoch ------------------------------------------
    def run_epoch(self, epoch=1):
        N = getattr(self.ctx, "N", 0)
        c = Chunk(0, 0, 0, N, 0, FORWARD, epoch, True)
        if self.gnn_type == GCN and self.numLayers < 2:
            # engine/utils.cpp:707-727: a one-layer GCN chunk never satisfies isLastLayer (host/engine.cpp refuses too)
            raise DoryError("GCN needs at least 2 layers: the reference's layer state machine has no epoch boundary otherwise")
        if self.gnn_type == GCN:
            while True:
                self.aggregateGCN(c)                       # GA
                if c.dir == BACKWARD and self._transform_first(c.layer):
                    self.ctx.weight_update(c.layer)        # transform-first: this aggregation produced dW_l
                c = self.applyVertexGCN(c)                 # AV (+ NNRecvCallbackGCN below)
                if self.isLastLayer(c):
                    if self._transform_first(0):           # dW0 = X^T (A^T g0): see include/dorylus_hip.h
                        self.scatterGCN(c)
                        self.aggregateGCN(c)
                        self.ctx.weight_update(0)
                    break
                if c.dir == FORWARD:
                    c = self.incLayerGCN(c)                # resource_comm.cpp:44-46
                self.scatterGCN(c)                         # SC
                # AE: applyEdgeGCN is a no-op (gcn_ops.cpp:364-366)
        else:
            while True:
                c.vertex = True
                c = self.applyVertexGAT(c)                 # AV
                if self.isLastLayer(c):
                    break
                if c.dir == FORWARD:
                    c = self.incLayerGAT(c)                # resource_comm.cpp:81-83
                self.scatterGAT(c)                         # SC
                self.applyEdgeGAT(c)                       # AE
                self.aggregateGAT(c)                       # GA
                if c.dir == FORWARD and c.layer == self.numLayers:   # pipeline.cpp:203-213
                    self.ctx.predict_gat(c.layer)
                    c.dir = BACKWARD
                    self.scatterGAT(c)
                    self.applyEdgeGAT(c)
                    self.aggregateGAT(c)
        return self.incLayerGCN(c) if self.gnn_type == GCN else self.incLayerGAT(c)


class NativeEngine:
    """The C++ Engine mirror (host/engine.cpp, include/dorylus_host.h): whole epochs run
    inside the library, no Python between stages."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        h = C.c_void_p()
        rc = ctx.lib.dory_engine_create(ctx.h, C.byref(h))
        if rc != 0:
            raise DoryError(f"dory_engine_create failed ({rc}): {ctx.lib.dory_last_error(ctx.h).decode()}")
        self.h = h

    def run(self, epochs):
        ms = np.zeros(epochs, np.float64)
        rc = self.ctx.lib.dory_engine_run(self.h, epochs, ms.ctypes.data)
        if rc != 0:
            raise DoryError(f"dory_engine_run failed ({rc}): {self.ctx.lib.dory_last_error(self.ctx.h).decode()}")
        return ms

    def report(self):
        buf = C.create_string_buffer(2048)
        self.ctx.lib.dory_engine_report(self.h, buf, 2048)
        return buf.value.decode()

    def close(self):
        if self.h:
            self.ctx.lib.dory_engine_destroy(self.h)
            self.h = None
