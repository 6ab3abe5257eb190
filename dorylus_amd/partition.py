"""ctypes front-end of the host-side partition builder / graph.<id>.bin IO
(include/dorylus_host.h; C++ in host/partition.cpp)."""
import ctypes as C

import numpy as np

from ._lib import DoryError, load


class _View(C.Structure):
    _fields_ = [("local_vtx_cnt", C.c_uint32), ("global_vtx_cnt", C.c_uint32),
                ("src_ghost_cnt", C.c_uint32), ("dst_ghost_cnt", C.c_uint32), ("num_nodes", C.c_uint32),
                ("local_in_edge_cnt", C.c_uint64), ("local_out_edge_cnt", C.c_uint64),
                ("global_edge_cnt", C.c_uint64),
                ("local_to_global", C.c_void_p), ("norms", C.c_void_p), ("src_ghosts", C.c_void_p),
                ("dst_ghosts", C.c_void_p), ("fwd_counts", C.c_void_p), ("fwd_lists", C.c_void_p),
                ("bwd_counts", C.c_void_p), ("bwd_lists", C.c_void_p), ("column_ptrs", C.c_void_p),
                ("row_idxs", C.c_void_p), ("csc_values", C.c_void_p), ("row_ptrs", C.c_void_p),
                ("column_idxs", C.c_void_p), ("csr_values", C.c_void_p)]


def _arr(ptr, n, dt):
    if n == 0 or not ptr:
        return np.zeros(0, dt)
    buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt, count=n)


class Partition:
    """One partition on the host: every field of graph.<id>.bin."""

    def __init__(self, handle, lib):
        self.h, self.lib = handle, lib

    @staticmethod
    def _ck(lib, rc):
        if rc != 0:
            raise DoryError(f"dorylus_host error {rc}: {lib.dory_host_last_error().decode()}")

    @classmethod
    def build(cls, src, dst, parts, node_id, num_nodes, undirected=False, lib=None):
        lib = lib or load()
        s = np.ascontiguousarray(src, np.uint32)
        d = np.ascontiguousarray(dst, np.uint32)
        p = np.ascontiguousarray(parts, np.int32)
        h = C.c_void_p()
        cls._ck(lib, lib.dory_partition_build(s.ctypes.data, d.ctypes.data, s.size, p.ctypes.data, p.size,
                                               node_id, num_nodes, int(undirected), C.byref(h)))
        return cls(h, lib)

    @classmethod
    def build_from_files(cls, dataset_dir, node_id, num_nodes, undirected=False, lib=None):
        lib = lib or load()
        h = C.c_void_p()
        cls._ck(lib, lib.dory_partition_build_from_files(dataset_dir.encode(), node_id, num_nodes,
                                                          int(undirected), C.byref(h)))
        return cls(h, lib)

    @classmethod
    def load(cls, path, lib=None):
        lib = lib or load()
        h = C.c_void_p()
        cls._ck(lib, lib.dory_partition_load(path.encode(), C.byref(h)))
        return cls(h, lib)

    def save(self, path):
        self._ck(self.lib, self.lib.dory_partition_save(self.h, path.encode()))

    def close(self):
        if self.h:
            self.lib.dory_partition_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def view(self):
        """dict of numpy views (valid while this object lives), keys as oracle/partition_oracle.py."""
        v = _View()
        self._ck(self.lib, self.lib.dory_partition_get(self.h, C.byref(v)))
        N, P = v.local_vtx_cnt, v.num_nodes
        fc = _arr(v.fwd_counts, P, np.uint32)
        bc = _arr(v.bwd_counts, P, np.uint32)
        fl = _arr(v.fwd_lists, int(fc.sum()), np.uint32)
        bl = _arr(v.bwd_lists, int(bc.sum()), np.uint32)
        fo = np.concatenate([[0], np.cumsum(fc)]).astype(np.int64)
        bo = np.concatenate([[0], np.cumsum(bc)]).astype(np.int64)
        return dict(
            localVtxCnt=N, globalVtxCnt=v.global_vtx_cnt, srcGhostCnt=v.src_ghost_cnt,
            dstGhostCnt=v.dst_ghost_cnt, numNodes=P, localInEdgeCnt=v.local_in_edge_cnt,
            localOutEdgeCnt=v.local_out_edge_cnt, globalEdgeCnt=v.global_edge_cnt,
            localToGlobal=_arr(v.local_to_global, N, np.uint32), norm=_arr(v.norms, N, np.float32),
            srcGhost=_arr(v.src_ghosts, v.src_ghost_cnt, np.uint32),
            dstGhost=_arr(v.dst_ghosts, v.dst_ghost_cnt, np.uint32),
            fwdLists=[fl[fo[i]:fo[i + 1]] for i in range(P)], bwdLists=[bl[bo[i]:bo[i + 1]] for i in range(P)],
            colPtr=_arr(v.column_ptrs, N + 1, np.uint64), rowIdx=_arr(v.row_idxs, v.local_in_edge_cnt, np.uint32),
            cscVal=_arr(v.csc_values, v.local_in_edge_cnt, np.float32),
            rowPtr=_arr(v.row_ptrs, N + 1, np.uint64), colIdx=_arr(v.column_idxs, v.local_out_edge_cnt, np.uint32),
            csrVal=_arr(v.csr_values, v.local_out_edge_cnt, np.float32))

    def recv_plan(self, parts, direction):
        """per-peer lists of ghost slots that peer's rows land in (host/partition.cpp)."""
        v = self.view()
        P = int(v["numNodes"])
        G = int(v["srcGhostCnt"] if direction == 0 else v["dstGhostCnt"])
        p = np.ascontiguousarray(parts, np.int32)
        cnt = np.zeros(P, np.uint32)
        slots = np.zeros(G + 1, np.uint32)
        self._ck(self.lib, self.lib.dory_partition_recv_plan(self.h, p.ctypes.data, direction,
                                                              cnt.ctypes.data, slots.ctypes.data))
        off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        return [slots[off[i]:off[i + 1]].copy() for i in range(P)]

    def upload(self, ctx, parts=None):
        """dory_graph_upload + both halo plans (when the .parts vector is given)."""
        p = None if parts is None else np.ascontiguousarray(parts, np.int32)
        rc = self.lib.dory_partition_upload(ctx.h, self.h, p.ctypes.data if p is not None else None)
        if rc != 0:
            raise DoryError(f"partition_upload failed ({rc}): {self.lib.dory_last_error(ctx.h).decode()} "
                            f"{self.lib.dory_host_last_error().decode()}")
        ctx.N = int(self.view()["localVtxCnt"])


def read_layer_config(path, lib=None):
    """run/<dataset>.config -> list of layer widths (Engine::readLayerConfigFile)."""
    lib = lib or load()
    dims = np.zeros(64, np.uint32)
    n = C.c_uint32()
    rc = lib.dory_read_layer_config(path.encode(), dims.ctypes.data, 64, C.byref(n))
    if rc != 0:
        raise DoryError(f"read_layer_config failed ({rc}): {lib.dory_formats_last_error().decode()}")
    return [int(x) for x in dims[:n.value]]


def read_features(path, part, dim, node_id=0, cache_dir=None):
    """features.bsnap -> (local N x F, ghost Gsrc x F) for this partition (Engine::readFeaturesFile)."""
    v = part.view()
    local = np.zeros((int(v["localVtxCnt"]), dim), np.float32)
    ghost = np.zeros((int(v["srcGhostCnt"]), dim), np.float32)
    gp = ghost.ctypes.data if ghost.size else None
    rc = part.lib.dory_read_features(path.encode(), part.h, dim, node_id,
                                     cache_dir.encode() if cache_dir else None, local.ctypes.data, gp)
    if rc != 0:
        raise DoryError(f"read_features failed ({rc}): {part.lib.dory_formats_last_error().decode()}")
    return local, ghost


def read_labels(path, part, kinds):
    v = part.view()
    lab = np.zeros(int(v["localVtxCnt"]), np.uint32)
    rc = part.lib.dory_read_labels(path.encode(), part.h, kinds, lab.ctypes.data)
    if rc != 0:
        raise DoryError(f"read_labels failed ({rc}): {part.lib.dory_formats_last_error().decode()}")
    return lab
