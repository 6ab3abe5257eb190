"""Epoch/layer/direction state machine and stage order of the Engine mirror, no GPU.

Expected sequences are restated here from the reference, independently of both mirrors:
  * scheduler -> GA (GCN) or AV (GAT): ops/pipeline.cpp:170-176
  * GA -> AV, or for GAT at layer == numLayers in forward: predictGAT, dir := BACKWARD, -> SC
    (gatherWorkFunc, pipeline.cpp:183-219)
  * AV forward: NNCompute(chunk); AV backward: NNCompute(incLayer(chunk)) (gcn_ops.cpp:194-202, gat_ops.cpp:267-275)
  * after AV: last layer -> next epoch; forward -> incLayer then SC; backward -> SC unchanged
    (resource_comm.cpp:17-51, 53-90); after AE (GAT): GA (:54-57)
  * SC -> AE, and applyEdgeGCN hands straight to GA (pipeline.cpp:262-342, gcn_ops.cpp:364-366)
  * incLayerGCN/GAT, isLastLayer: engine/utils.cpp:707-753
  * weight updates leave after vtxNNBackward and after the last forward layer of GCN
    (CPU_comm.cpp:131,147,178)
"""
import ctypes as C

import numpy as np
import pytest

import dorylus_amd as da
from dorylus_amd._lib import load
from dorylus_amd.engine import Chunk, Engine

F, B = da.FORWARD, da.BACKWARD


def ref_inc_gcn(layer, d, epoch, L):            # engine/utils.cpp:707-727
    if d == F:
        layer += 1
        if layer == L:
            d, layer = B, layer - 1
    elif layer == 0:
        d, epoch = F, epoch + 1
    else:
        layer -= 1
    return layer, d, epoch


def ref_inc_gat(layer, d, epoch, L):            # engine/utils.cpp:729-748
    if d == F:
        layer += 1
    elif layer == 0:
        d, epoch = F, epoch + 1
    else:
        layer -= 1
    return layer, d, epoch


def expected_trace(gnn, L):
    """walk one chunk through the reference's queues by the rules in the module docstring"""
    name = lambda st, l, d: f"{st}{l}{'F' if d == F else 'B'}"
    out, layer, d = [], 0, F
    if gnn == da.GCN:
        while True:
            out.append(name("GA", layer, d))
            if d == F:
                out.append(name("AV", layer, d))
                if layer == L - 1:
                    out.append(f"WU{layer}")
                layer, d, _ = ref_inc_gcn(layer, d, 1, L)          # callback: forward -> inc, then SC
            else:
                layer, d, _ = ref_inc_gcn(layer, d, 1, L)          # AVB: inc first
                out.append(name("AV", layer, d))
                out.append(f"WU{layer}")
                if layer == 0:
                    return out
            out.append(name("SC", layer, d))
    while True:
        if d == F:
            out.append(name("AV", layer, d))
            layer, d, _ = ref_inc_gat(layer, d, 1, L)
        else:
            layer, d, _ = ref_inc_gat(layer, d, 1, L)
            out.append(name("AV", layer, d))
            out.append(f"WU{layer}")
            if layer == 0:
                return out
        out += [name("SC", layer, d), name("AE", layer, d), name("GA", layer, d)]
        if d == F and layer == L:
            out.append(f"PR{layer}")
            d = B
            out += [name("SC", layer, d), name("AE", layer, d), name("GA", layer, d)]


class Recorder:
    """stands in for the Context: records the C-ABI calls the Python mirror issues"""
    N = 0

    def __init__(self):
        self.calls = []

    def _rec(self, st):
        return lambda layer, d=None: self.calls.append(f"{st}{layer}" + ("" if d is None else "F" if d == F else "B"))

    def __getattr__(self, nm):
        table = {"aggregate": "GA", "apply_vertex": "AV", "apply_edge": "AE", "halo_exchange": "SC",
                 "weight_update": "WU", "predict_gat": "PR"}
        if nm in table:
            return self._rec(table[nm])
        raise AttributeError(nm)


def test_known_answer_two_layer_sequences():
    assert expected_trace(da.GCN, 2) == "GA0F AV0F SC1F GA1F AV1F WU1 SC1B GA1B AV0B WU0".split()
    assert expected_trace(da.GAT, 2) == ("AV0F SC1F AE1F GA1F AV1F SC2F AE2F GA2F PR2 SC2B AE2B GA2B "
                                        "AV1B WU1 SC1B AE1B GA1B AV0B WU0").split()


@pytest.mark.parametrize("gnn", [da.GCN, da.GAT])
@pytest.mark.parametrize("L", [1, 2, 3, 5])
def test_stage_order_cpp_and_python_mirrors(gnn, L):
    lib = load()
    if gnn == da.GCN and L == 1:
        # the reference's GCN state machine has no epoch boundary with one layer (the merged last
        # forward layer falls back into a forward chunk): the engine refuses instead of spinning
        assert lib.dory_engine_trace_epoch(gnn, L, C.create_string_buffer(64), 64) != 0
        with pytest.raises(da.DoryError):
            Engine(Recorder(), gnn, L).run_epoch(1)
        return
    buf = C.create_string_buffer(4096)
    assert lib.dory_engine_trace_epoch(gnn, L, buf, 4096) == 0
    cpp = buf.value.decode().split()
    rec = Recorder()
    Engine(rec, gnn, L).run_epoch(1)
    want = expected_trace(gnn, L)
    assert cpp == want
    assert rec.calls == want


def test_trace_rejects_bad_arguments():
    lib = load()
    buf = C.create_string_buffer(8)
    assert lib.dory_engine_trace_epoch(da.GCN, 0, buf, 8) != 0
    assert lib.dory_engine_trace_epoch(7, 2, buf, 8) != 0
    assert lib.dory_engine_trace_epoch(da.GCN, 3, buf, 8) != 0          # buffer too small


class CChunk(C.Structure):
    _fields_ = [("localId", C.c_uint32), ("globalId", C.c_uint32), ("lowBound", C.c_uint32), ("upBound", C.c_uint32),
                ("layer", C.c_uint32), ("dir", C.c_int32), ("epoch", C.c_uint32), ("vertex", C.c_uint8)]


@pytest.mark.parametrize("gnn", [da.GCN, da.GAT])
@pytest.mark.parametrize("L", [1, 2, 4])
def test_inc_layer_every_state(gnn, L):
    lib = load()
    ref = ref_inc_gcn if gnn == da.GCN else ref_inc_gat
    eng = Engine(Recorder(), gnn, L)
    for layer in range(L + 1):
        for d in (F, B):
            for vertex in (0, 1):
                if gnn == da.GCN and d == F and layer >= L:
                    continue                       # GCN never holds a forward chunk past the last layer
                cin = CChunk(3, 4, 5, 99, layer, d, 7, vertex)
                cout = CChunk()
                assert lib.dory_chunk_inc_layer(gnn, L, C.byref(cin), C.byref(cout)) == 0
                wl, wd, we = ref(layer, d, 7, L)
                assert (cout.layer, cout.dir, cout.epoch) == (wl, wd, we)
                assert (cout.localId, cout.globalId, cout.lowBound, cout.upBound) == (3, 4, 5, 99)
                # GAT resets `vertex` when it re-enters the forward pass (engine/utils.cpp:741)
                want_vertex = 1 if (gnn == da.GAT and d == B and layer == 0) else vertex
                assert cout.vertex == want_vertex
                py = (eng.incLayerGCN if gnn == da.GCN else eng.incLayerGAT)(Chunk(3, 4, 5, 99, layer, d, 7, bool(vertex)))
                assert (py.layer, py.dir, py.epoch, int(py.vertex)) == (wl, wd, we, want_vertex)
