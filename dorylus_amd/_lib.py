"""ctypes binding of the C-ABI in include/dorylus_hip.h.

There is no CPU fallback: a missing library or a missing gfx950 device raises.
`import torch` (if the caller uses it) must happen before this module loads the
library so that both share one HIP runtime (same SONAME libamdhip64.so.7).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DORY_LIB_PATH") or os.path.join(_HERE, "libdorylus_hip.so")   # DORY_LIB_PATH: A/B runs against another build

# every symbol include/dorylus_hip.h declares (checked by tests/test_abi_symbols.py)
SYMBOLS = [
    "dory_create", "dory_destroy", "dory_last_error", "dory_set_streams", "dory_sync",
    "dory_configure", "dory_graph_upload", "dory_preallocate", "dory_tensor_info",
    "dory_tensor_upload", "dory_tensor_download", "dory_tensor_fill_uniform", "dory_labels_upload",
    "dory_weight_set", "dory_weight_get", "dory_weight_grad_get", "dory_weight_grad_set", "dory_weights_init_xavier",
    "dory_aggregate", "dory_apply_vertex", "dory_apply_edge", "dory_predict_gat", "dory_train_stat", "dory_train_stat_global",
    "dory_halo_plan", "dory_comm_unique_id", "dory_comm_init", "dory_halo_exchange", "dory_halo_pack",
    "dory_halo_unpack", "dory_halo_pack_tensor", "dory_halo_unpack_tensor", "dory_adam_config", "dory_weight_update", "dory_timing_enable",
    "dory_timing_get", "dory_timing_reset", "dory_set_option", "dory_get_option", "dory_ctx_describe",
    "dory_epoch_graph_begin", "dory_epoch_graph_end", "dory_epoch_graph_launch", "dory_epoch_graph_drop",
    "dory_gatmh_heads", "dory_transform_first_active", "dory_transform_first_layer",
    "dory_comm_set_host_transport", "dory_comm_init_local", "dory_debug_occupy_cus",
]

# host transport callbacks (include/dorylus_hip.h)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                           C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint32)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_uint64)

FORWARD, BACKWARD = 0, 1
GCN, GAT, GATMH = 0, 1, 2


class DoryError(RuntimeError):
    pass


def load():
    if not os.path.exists(LIB_PATH):
        raise DoryError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, u32, u64, i32, f32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_float
    cp = C.c_char_p
    sig = {
        "dory_create": [i32, C.POINTER(vp)],
        "dory_destroy": [vp],
        "dory_set_streams": [vp, vp, vp],
        "dory_debug_occupy_cus": [vp, u32, u64],
        "dory_sync": [vp],
        "dory_configure": [vp, i32, u32, vp, u32, u32, u32],
        "dory_graph_upload": [vp, u32, u32, u32, u64, vp, vp, vp, u64, vp, vp, vp, vp],
        "dory_preallocate": [vp],
        "dory_tensor_info": [vp, u32, cp, C.POINTER(u64), C.POINTER(u32), C.POINTER(u32), C.POINTER(vp)],
        "dory_tensor_upload": [vp, u32, cp, vp],
        "dory_tensor_download": [vp, u32, cp, vp],
        "dory_tensor_fill_uniform": [vp, u32, cp, u64, f32, f32, vp],
        "dory_labels_upload": [vp, vp],
        "dory_weight_set": [vp, u32, cp, vp],
        "dory_weight_get": [vp, u32, cp, vp],
        "dory_weight_grad_get": [vp, u32, cp, vp],
        "dory_weight_grad_set": [vp, u32, cp, vp],
        "dory_weights_init_xavier": [vp],
        "dory_aggregate": [vp, u32, i32],
        "dory_apply_vertex": [vp, u32, i32],
        "dory_apply_edge": [vp, u32, i32],
        "dory_predict_gat": [vp, u32],
        "dory_train_stat": [vp, C.POINTER(f32), C.POINTER(f32), C.POINTER(u32)],
        "dory_train_stat_global": [vp, C.POINTER(f32), C.POINTER(f32), C.POINTER(u32)],
        "dory_halo_plan": [vp, i32, vp, vp, vp, vp],
        "dory_comm_unique_id": [vp],
        "dory_comm_init": [vp, vp, i32, i32],
        "dory_halo_exchange": [vp, u32, i32],
        "dory_halo_pack": [vp, u32, i32, vp],
        "dory_halo_unpack": [vp, u32, i32, vp],
        "dory_halo_pack_tensor": [vp, u32, cp, i32, vp],
        "dory_halo_unpack_tensor": [vp, u32, cp, i32, vp],
        "dory_adam_config": [vp, f32],
        "dory_weight_update": [vp, u32],
        "dory_timing_enable": [vp, i32],
        "dory_timing_get": [vp, cp, C.POINTER(C.c_double), C.POINTER(u64)],
        "dory_timing_reset": [vp],
        "dory_set_option": [vp, cp, C.c_int64],
        "dory_get_option": [vp, cp, C.POINTER(C.c_int64)],
        "dory_transform_first_active": [vp],
        "dory_transform_first_layer": [vp, u32],
        "dory_epoch_graph_begin": [vp], "dory_epoch_graph_end": [vp], "dory_epoch_graph_launch": [vp, u32],
        "dory_epoch_graph_drop": [vp],
        "dory_ctx_describe": [vp, C.POINTER(i32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)],
        "dory_gatmh_heads": [vp, vp],
        "dory_comm_set_host_transport": [vp, ALLTOALLV_FN, ALLREDUCE_FN, vp],
        "dory_comm_init_local": [vp, C.c_uint32],
        # include/dorylus_host.h
        "dory_partition_build": [vp, vp, u64, vp, u32, u32, u32, i32, C.POINTER(vp)],
        "dory_partition_build_from_files": [cp, u32, u32, i32, C.POINTER(vp)],
        "dory_partition_load": [cp, C.POINTER(vp)],
        "dory_partition_save": [vp, cp],
        "dory_partition_free": [vp],
        "dory_partition_get": [vp, vp],
        "dory_partition_upload": [vp, vp, vp],
        "dory_partition_recv_plan": [vp, vp, i32, vp, vp],
        "dory_read_layer_config": [cp, vp, u32, C.POINTER(u32)],
        "dory_read_features": [cp, vp, u32, u32, cp, vp, vp],
        "dory_read_labels": [cp, vp, u32, vp],
        "dory_engine_create": [vp, C.POINTER(vp)],
        "dory_engine_destroy": [vp],
        "dory_engine_run": [vp, u32, vp],
        "dory_engine_nn_compute": [vp, vp],
        "dory_engine_inc_layer": [vp, vp, vp],
        "dory_chunk_inc_layer": [i32, u32, vp, vp],
        "dory_engine_trace_epoch": [i32, u32, C.c_char_p, C.c_size_t],
        "dory_engine_is_last_layer": [vp, vp],
        "dory_engine_report": [vp, cp, C.c_size_t],
        "dory_sweep_deal": [u32, u32, u32, vp, vp, vp],
        "dory_sweep_deal_weighted": [u32, vp, u32, u32, u32, vp, vp, vp],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i32
    lib.dory_last_error.argtypes = [vp]
    lib.dory_last_error.restype = cp
    lib.dory_host_last_error.argtypes = []
    lib.dory_host_last_error.restype = cp
    lib.dory_formats_last_error.argtypes = []
    lib.dory_formats_last_error.restype = cp
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """One GPU / one graph partition ("node" in the reference).  Thin, 1:1 with the C-ABI."""

    def __init__(self, device=0, lib=None):
        self.lib = lib or load()
        h = C.c_void_p()
        rc = self.lib.dory_create(device, C.byref(h))
        if rc != 0:
            raise DoryError(f"dory_create failed ({rc}): {self.lib.dory_last_error(None).decode()}")
        self.h = h
        self._keep = []

    def _ck(self, rc):
        if rc != 0:
            raise DoryError(f"dorylus_hip error {rc}: {self.lib.dory_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.lib.dory_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- setup -----------------------------------------------------------------
    def configure(self, gnn, dims, global_vtx_cnt, node_id=0, num_nodes=1):
        d = np.ascontiguousarray(dims, dtype=np.uint32)
        self.dims = [int(x) for x in d]
        self.L = len(d) - 1
        self._ck(self.lib.dory_configure(self.h, gnn, self.L, _ptr(d), global_vtx_cnt, node_id, num_nodes))

    def graph_upload(self, g):
        """g: dict with the fields of graph.<id>.bin."""
        cp = np.ascontiguousarray(g["colPtr"], np.uint64)
        ri = np.ascontiguousarray(g["rowIdx"], np.uint32)
        cv = np.ascontiguousarray(g["cscVal"], np.float32)
        rp = np.ascontiguousarray(g["rowPtr"], np.uint64)
        ci = np.ascontiguousarray(g["colIdx"], np.uint32)
        rv = np.ascontiguousarray(g["csrVal"], np.float32)
        nm = np.ascontiguousarray(g["norm"], np.float32)
        self.N = int(g["localVtxCnt"])
        self._ck(self.lib.dory_graph_upload(
            self.h, self.N, int(g["srcGhostCnt"]), int(g["dstGhostCnt"]), int(ri.size), _ptr(cp),
            _ptr(ri), _ptr(cv), int(ci.size), _ptr(rp), _ptr(ci), _ptr(rv), _ptr(nm)))

    def gatmh_heads(self, heads):
        h = np.ascontiguousarray(heads, np.uint32)
        assert h.size == self.L
        self.heads = [int(x) for x in h]
        self._ck(self.lib.dory_gatmh_heads(self.h, _ptr(h)))

    def preallocate(self):
        self._ck(self.lib.dory_preallocate(self.h))

    # -- tensors ------------------------------------------------------------------
    def info(self, layer, name):
        rows, cols, ld, p = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_void_p()
        self._ck(self.lib.dory_tensor_info(self.h, layer, name.encode(), C.byref(rows), C.byref(cols),
                                           C.byref(ld), C.byref(p)))
        return rows.value, cols.value, ld.value, p.value

    def upload(self, layer, name, a):
        rows, cols, _, _ = self.info(layer, name)
        a = np.ascontiguousarray(a, dtype=np.float32)
        if a.size != rows * cols:
            raise DoryError(f"upload {name}@{layer}: got {a.shape}, tensor is {rows}x{cols}")
        self._ck(self.lib.dory_tensor_upload(self.h, layer, name.encode(), _ptr(a)))

    def download(self, layer, name):
        rows, cols, _, _ = self.info(layer, name)
        a = np.empty((rows, cols), np.float32)
        self._ck(self.lib.dory_tensor_download(self.h, layer, name.encode(), _ptr(a)))
        return a

    def fill_uniform(self, layer, name, seed, lo=-1.0, hi=1.0, row_ids=None):
        ids = None if row_ids is None else np.ascontiguousarray(row_ids, np.uint32)
        self._ck(self.lib.dory_tensor_fill_uniform(self.h, layer, name.encode(), seed, lo, hi, _ptr(ids)))

    def labels_upload(self, labels):
        l = np.ascontiguousarray(labels, np.uint32)
        self._ck(self.lib.dory_labels_upload(self.h, _ptr(l)))

    # -- weights ---------------------------------------------------------------------
    def _wshape(self, layer, name):
        heads = getattr(self, "heads", None)
        if heads is not None:   # multi-head GAT extension: the last layer keeps K_out copies of the class width
            zw = self.dims[layer + 1] * (heads[layer] if layer == self.L - 1 else 1)
            return (self.dims[layer], zw) if name == "w" else (zw, 1)
        return (self.dims[layer], self.dims[layer + 1]) if name == "w" else (self.dims[layer + 1], 1)

    def weight_set(self, layer, name, w):
        w = np.ascontiguousarray(w, np.float32)
        assert w.size == int(np.prod(self._wshape(layer, name)))
        self._ck(self.lib.dory_weight_set(self.h, layer, name.encode(), _ptr(w)))

    def weight_get(self, layer, name="w"):
        w = np.empty(self._wshape(layer, name), np.float32)
        self._ck(self.lib.dory_weight_get(self.h, layer, name.encode(), _ptr(w)))
        return w

    def weight_grad_get(self, layer, name="w"):
        w = np.empty(self._wshape(layer, name), np.float32)
        self._ck(self.lib.dory_weight_grad_get(self.h, layer, name.encode(), _ptr(w)))
        return w

    def weight_grad_set(self, layer, grad, name="w"):
        g = np.ascontiguousarray(grad, np.float32)
        assert g.shape == self._wshape(layer, name)
        self._ck(self.lib.dory_weight_grad_set(self.h, layer, name.encode(), _ptr(g)))

    def weights_init_xavier(self):
        self._ck(self.lib.dory_weights_init_xavier(self.h))

    # -- hot path ------------------------------------------------------------------------
    def aggregate(self, layer, direction):
        self._ck(self.lib.dory_aggregate(self.h, layer, direction))

    def apply_vertex(self, layer, direction):
        self._ck(self.lib.dory_apply_vertex(self.h, layer, direction))

    def apply_edge(self, layer, direction):
        self._ck(self.lib.dory_apply_edge(self.h, layer, direction))

    def predict_gat(self, layer):
        self._ck(self.lib.dory_predict_gat(self.h, layer))

    def train_stat(self):
        a, l, n = C.c_float(), C.c_float(), C.c_uint32()
        self._ck(self.lib.dory_train_stat(self.h, C.byref(a), C.byref(l), C.byref(n)))
        return a.value, l.value, n.value

    def train_stat_global(self):
        """the same summed over all partitions (a collective: every rank calls it)"""
        a, l, n = C.c_float(), C.c_float(), C.c_uint32()
        self._ck(self.lib.dory_train_stat_global(self.h, C.byref(a), C.byref(l), C.byref(n)))
        return a.value, l.value, n.value

    # -- halo / comm ------------------------------------------------------------------------
    def halo_plan(self, direction, send_lists, recv_slot_lists):
        sc = np.array([len(x) for x in send_lists], np.uint32)
        rc = np.array([len(x) for x in recv_slot_lists], np.uint32)
        sl = np.ascontiguousarray(np.concatenate([np.asarray(x, np.uint32) for x in send_lists] + [np.zeros(0, np.uint32)]), np.uint32)
        rs = np.ascontiguousarray(np.concatenate([np.asarray(x, np.uint32) for x in recv_slot_lists] + [np.zeros(0, np.uint32)]), np.uint32)
        self._ck(self.lib.dory_halo_plan(self.h, direction, _ptr(sc), _ptr(sl), _ptr(rc), _ptr(rs)))

    def comm_unique_id(self):
        buf = np.zeros(128, np.uint8)
        rc = self.lib.dory_comm_unique_id(_ptr(buf))
        if rc != 0:
            raise DoryError("ncclGetUniqueId failed")
        return buf

    def comm_init(self, id128, rank, nranks):
        b = np.ascontiguousarray(id128, np.uint8)
        self._ck(self.lib.dory_comm_init(self.h, _ptr(b), rank, nranks))

    @staticmethod
    def comm_init_local(ctxs):
        """in-process device transport: the contexts (rank i = ctxs[i]) of this process become each other's peers -- rows
        travel device -> device on the sender's comm stream, ordered by cross-context events (dory_comm_init_local).
        Drive each rank from its own thread afterwards (ctypes calls release the GIL)."""
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        rc = ctxs[0].lib.dory_comm_init_local(arr, len(ctxs))
        if rc != 0:
            msgs = [c.lib.dory_last_error(c.h).decode() for c in ctxs]
            raise DoryError(f"dory_comm_init_local failed ({rc}): " + "; ".join(m for m in msgs if m))

    def set_host_transport(self, alltoallv, allreduce):
        """alltoallv(send: np.ndarray, send_counts, send_offsets, recv: np.ndarray, recv_counts, recv_offsets) and
        allreduce(buf: np.ndarray) move host floats between the ranks (e.g. over gloo); everything else of the
        multi-rank epoch stays in the library.  (None, None) returns to RCCL."""
        if alltoallv is None:
            self._tx = None
            self._ck(self.lib.dory_comm_set_host_transport(self.h, ALLTOALLV_FN(0), ALLREDUCE_FN(0), None))
            return

        def a2a(user, send, sc, so, recv, rc, ro, P):
            try:
                scn, son = np.ctypeslib.as_array(sc, (P,)).copy(), np.ctypeslib.as_array(so, (P,)).copy()
                rcn, ron = np.ctypeslib.as_array(rc, (P,)).copy(), np.ctypeslib.as_array(ro, (P,)).copy()
                ns, nr = int((son + scn).max()), int((ron + rcn).max())
                sa = np.ctypeslib.as_array(send, (max(ns, 1),))
                ra = np.ctypeslib.as_array(recv, (max(nr, 1),))
                alltoallv(sa, scn, son, ra, rcn, ron)
                return 0
            except Exception:       # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        def ar(user, buf, n):
            try:
                allreduce(np.ctypeslib.as_array(buf, (int(n),)))
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        self._tx = (ALLTOALLV_FN(a2a), ALLREDUCE_FN(ar))      # keep the thunks alive
        self._ck(self.lib.dory_comm_set_host_transport(self.h, self._tx[0], self._tx[1], None))

    def halo_exchange(self, layer, direction):
        self._ck(self.lib.dory_halo_exchange(self.h, layer, direction))

    def halo_pack(self, layer, direction, dev_ptr):
        self._ck(self.lib.dory_halo_pack(self.h, layer, direction, C.c_void_p(dev_ptr)))

    def halo_unpack(self, layer, direction, dev_ptr):
        self._ck(self.lib.dory_halo_unpack(self.h, layer, direction, C.c_void_p(dev_ptr)))

    def halo_pack_tensor(self, layer, name, direction, dev_ptr):
        self._ck(self.lib.dory_halo_pack_tensor(self.h, layer, name.encode(), direction, C.c_void_p(dev_ptr)))

    def halo_unpack_tensor(self, layer, name, direction, dev_ptr):
        self._ck(self.lib.dory_halo_unpack_tensor(self.h, layer, name.encode(), direction, C.c_void_p(dev_ptr)))

    # -- optimiser ---------------------------------------------------------------------------
    def adam_config(self, lr):
        self._ck(self.lib.dory_adam_config(self.h, lr))

    def weight_update(self, layer):
        self._ck(self.lib.dory_weight_update(self.h, layer))

    # -- misc -----------------------------------------------------------------------------------
    def sync(self):
        self._ck(self.lib.dory_sync(self.h))

    def debug_occupy_cus(self, workgroups, usec):
        self._ck(self.lib.dory_debug_occupy_cus(self.h, workgroups, usec))

    def set_streams(self, compute=None, comm=None):
        self._ck(self.lib.dory_set_streams(self.h, C.c_void_p(compute or 0), C.c_void_p(comm or 0)))

    def timing_enable(self, on=True):
        self._ck(self.lib.dory_timing_enable(self.h, int(on)))

    def timing_get(self, family):
        ms, n = C.c_double(), C.c_uint64()
        self._ck(self.lib.dory_timing_get(self.h, family.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def timing_reset(self):
        self._ck(self.lib.dory_timing_reset(self.h))

    def set_option(self, key, value):
        self._ck(self.lib.dory_set_option(self.h, key.encode(), int(value)))

    def transform_first_active(self):
        return bool(self.lib.dory_transform_first_active(self.h))

    def transform_first_layer(self, layer):
        return bool(self.lib.dory_transform_first_layer(self.h, int(layer)))

    def get_option(self, key):
        v = C.c_int64(0)
        self._ck(self.lib.dory_get_option(self.h, key.encode(), C.byref(v)))
        return int(v.value)

    # epoch graph (hipGraph replay of one recorded epoch; single partition)
    def epoch_graph_begin(self):
        self._ck(self.lib.dory_epoch_graph_begin(self.h))

    def epoch_graph_end(self):
        self._ck(self.lib.dory_epoch_graph_end(self.h))

    def epoch_graph_launch(self, epochs=1):
        self._ck(self.lib.dory_epoch_graph_launch(self.h, int(epochs)))

    def epoch_graph_drop(self):
        self._ck(self.lib.dory_epoch_graph_drop(self.h))
