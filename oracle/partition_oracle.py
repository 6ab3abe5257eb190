"""partition_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the reference's partition builder
(`DataLoader::preprocess`, src/graph-server/graph/dataloader.cpp:225-330) and of
its `graph.<id>.bin` writer (`RawGraph::dump`, graph/graph.cpp:200-273).

Pinned bit-exact against the reference's own DataLoader compiled unmodified into
oracle/_ref/ref_preprocess (tests/test_oracle_partition.py; committed fixtures
under tests/golden/parts_*).  Only tests/, smoke() and bench.py's cpu_baseline
leg may import this.
"""
import math
import struct

import numpy as np


# ----------------------------------------------------------------------------
# file formats next to the hot path
# ----------------------------------------------------------------------------
def write_bsnap_edges(path, num_vertices, src, dst):
    """graph.bsnap.edges: BSHeaderType {int sizeOfVertexType; unsigned numVertices;
    unsigned long long numEdges} (graph/dataloader.hpp:11-15) + (u32 src,u32 dst)."""
    src = np.asarray(src, dtype=np.uint32)
    dst = np.asarray(dst, dtype=np.uint32)
    with open(path, "wb") as f:
        f.write(struct.pack("<iIQ", 4, int(num_vertices), int(src.size)))
        np.stack([src, dst], axis=1).astype("<u4").tofile(f)


def read_bsnap_edges(path):
    with open(path, "rb") as f:
        sz, nv, ne = struct.unpack("<iIQ", f.read(16))
        assert sz == 4
        e = np.fromfile(f, dtype="<u4").reshape(-1, 2)
    assert e.shape[0] == ne
    return nv, e[:, 0].copy(), e[:, 1].copy()


def write_parts(path, parts):
    """graph.bsnap.parts: one partition id per line (dataloader.cpp:53-87)."""
    with open(path, "w") as f:
        f.write("\n".join(str(int(p)) for p in parts))
        f.write("\n")


def write_features(path, feats):
    """features.bsnap: u32 numFeatures + rows (engine.hpp:30-32)."""
    feats = np.ascontiguousarray(feats, dtype="<f4")
    with open(path, "wb") as f:
        f.write(struct.pack("<I", feats.shape[1]))
        feats.tofile(f)


def write_labels(path, labels, kinds):
    """labels.bsnap: u32 labelKinds + u32 labels (engine.hpp:35-37)."""
    with open(path, "wb") as f:
        f.write(struct.pack("<I", int(kinds)))
        np.asarray(labels, dtype="<u4").tofile(f)


# ----------------------------------------------------------------------------
# DataLoader::preprocess restated
# ----------------------------------------------------------------------------
def _inv_sqrt_f32(deg):
    """float vtxNorm = std::pow(unsigned deg, -.5)  (dataloader.cpp:155-156):
    libm pow in double, narrowed to float."""
    deg = np.asarray(deg, dtype=np.int64)
    if deg.size == 0:
        return np.zeros(0, dtype=np.float32)
    uniq, inv = np.unique(deg, return_inverse=True)
    table = np.array([math.pow(float(d), -0.5) for d in uniq], dtype=np.float64)
    return table[inv].astype(np.float32).reshape(deg.shape)


def preprocess(src, dst, parts, node_id, num_nodes, undirected=False):
    """Returns a dict with every field of graph.<id>.bin."""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    parts = np.asarray(parts, dtype=np.int64)
    V = parts.size

    # readPartsFile (dataloader.cpp:53-87): local ids ascend with global id.
    l2g = np.nonzero(parts == node_id)[0]
    N = l2g.size
    g2l = np.full(V, -1, dtype=np.int64)
    g2l[l2g] = np.arange(N)

    # edge loop (dataloader.cpp:266-275): skip self loops, count records once.
    keep = src != dst
    fs, fd = src[keep], dst[keep]                 # file records
    global_edge_cnt = int(fs.size)
    if undirected:                                # processEdge(s,d); processEdge(d,s)
        ps = np.stack([fs, fd], axis=1).reshape(-1)
        pd = np.stack([fd, fs], axis=1).reshape(-1)
    else:
        ps, pd = fs, fd

    # processEdge (dataloader.cpp:94-146)
    out_m = parts[ps] == node_id                  # out-edge of a local vertex
    in_m = parts[pd] == node_id                   # in-edge of a local vertex
    o_from, o_to = ps[out_m], pd[out_m]
    i_from, i_to = ps[in_m], pd[in_m]
    o_remote = parts[o_to] != node_id
    i_remote = parts[i_from] != node_id

    out_ghost = np.unique(o_to[o_remote])         # std::map keys: ascending gvid
    in_ghost = np.unique(i_from[i_remote])

    # per-peer send lists (dataloader.cpp:277-297): ascending local id
    fwd_lists, bwd_lists = [], []
    for p in range(num_nodes):
        if p == node_id:
            fwd_lists.append(np.zeros(0, dtype=np.uint32))
            bwd_lists.append(np.zeros(0, dtype=np.uint32))
            continue
        m = o_remote & (parts[o_to] == p)
        fwd_lists.append(np.unique(g2l[o_from[m]]).astype(np.uint32))
        m = i_remote & (parts[i_from] == p)
        bwd_lists.append(np.unique(g2l[i_to[m]]).astype(np.uint32))

    # findGhostDegrees (dataloader.cpp:192-218): counts FILE records whose dst
    # is the ghost, for both ghost kinds (reverse records are not counted).
    indeg_file = np.bincount(fd, minlength=V)
    # local in-degree = number of in-edges stored on the vertex
    in_cnt = np.bincount(g2l[i_to], minlength=N)
    out_cnt = np.bincount(g2l[o_from], minlength=N)

    # setEdgeNormalizations (dataloader.cpp:153-185)
    vnorm = _inv_sqrt_f32(in_cnt + 1)             # per local vertex
    norm = (vnorm * vnorm).astype(np.float32)     # vertex.setNormFactor
    in_ghost_norm = _inv_sqrt_f32(indeg_file[in_ghost] + 1)
    out_ghost_norm = _inv_sqrt_f32(indeg_file[out_ghost] + 1)

    # CSC (graph/graph.hpp:167-190): per destination column, edge-file order
    lto = g2l[i_to]
    order = np.argsort(lto, kind="stable")
    colptr = np.zeros(N + 1, dtype=np.uint64)
    colptr[1:] = np.cumsum(in_cnt).astype(np.uint64)
    s_from, s_remote, s_to = i_from[order], i_remote[order], lto[order]
    gi = np.searchsorted(in_ghost, s_from)        # ghost rank (valid where remote)
    rowidx = np.where(s_remote, N + gi, g2l[s_from]).astype(np.uint32)
    gnorm = (in_ghost_norm[np.minimum(gi, in_ghost.size - 1)] if in_ghost.size
             else np.zeros(s_from.size, dtype=np.float32))
    lnorm = vnorm[np.where(s_remote, 0, g2l[s_from])] if N else gnorm
    src_norm = np.where(s_remote, gnorm, lnorm).astype(np.float32)
    csc_val = (src_norm * vnorm[s_to]).astype(np.float32)   # srcNorm * vtxNorm

    # CSR (graph/graph.hpp:192-215): per source row, edge-file order
    lfrom = g2l[o_from]
    order = np.argsort(lfrom, kind="stable")
    rowptr = np.zeros(N + 1, dtype=np.uint64)
    rowptr[1:] = np.cumsum(out_cnt).astype(np.uint64)
    t_to, t_remote, t_from = o_to[order], o_remote[order], lfrom[order]
    gi = np.searchsorted(out_ghost, t_to)
    colidx = np.where(t_remote, N + gi, g2l[t_to]).astype(np.uint32)
    gnorm = (out_ghost_norm[np.minimum(gi, out_ghost.size - 1)] if out_ghost.size
             else np.zeros(t_to.size, dtype=np.float32))
    lnorm = vnorm[np.where(t_remote, 0, g2l[t_to])] if N else gnorm
    dst_norm = np.where(t_remote, gnorm, lnorm).astype(np.float32)
    csr_val = (vnorm[t_from] * dst_norm).astype(np.float32)  # vtxNorm * dstNorm

    return dict(
        localVtxCnt=N, globalVtxCnt=V, srcGhostCnt=int(in_ghost.size),
        dstGhostCnt=int(out_ghost.size), localInEdgeCnt=int(i_to.size),
        localOutEdgeCnt=int(o_from.size), globalEdgeCnt=global_edge_cnt,
        localToGlobal=l2g.astype(np.uint32), norm=norm,
        srcGhost=in_ghost.astype(np.uint32), dstGhost=out_ghost.astype(np.uint32),
        numNodes=num_nodes, fwdLists=fwd_lists, bwdLists=bwd_lists,
        colPtr=colptr, rowIdx=rowidx, cscVal=csc_val,
        rowPtr=rowptr, colIdx=colidx, csrVal=csr_val)


def dump_bytes(g):
    """RawGraph::dump (graph/graph.cpp:200-273) -> bytes of graph.<id>.bin."""
    N = g["localVtxCnt"]
    out = [struct.pack("<IIII", N, g["globalVtxCnt"], g["srcGhostCnt"], g["dstGhostCnt"]),
           struct.pack("<QQQ", g["localInEdgeCnt"], g["localOutEdgeCnt"], g["globalEdgeCnt"]),
           g["localToGlobal"].astype("<u4").tobytes(),
           g["norm"].astype("<f4").tobytes()]
    for key in ("srcGhost", "dstGhost"):
        gv = g[key].astype("<u4")
        pairs = np.stack([gv, N + np.arange(gv.size, dtype=np.uint32)], axis=1)
        out.append(pairs.astype("<u4").tobytes())
    out.append(struct.pack("<I", g["numNodes"]))
    for lists in (g["fwdLists"], g["bwdLists"]):
        for l in lists:
            out.append(struct.pack("<I", l.size))
            out.append(l.astype("<u4").tobytes())
    out.append(struct.pack("<IQ", N, g["localInEdgeCnt"]))
    out.append(g["cscVal"].astype("<f4").tobytes())
    out.append(g["colPtr"].astype("<u8").tobytes())
    out.append(g["rowIdx"].astype("<u4").tobytes())
    out.append(struct.pack("<IQ", N, g["localOutEdgeCnt"]))
    out.append(g["csrVal"].astype("<f4").tobytes())
    out.append(g["rowPtr"].astype("<u8").tobytes())
    out.append(g["colIdx"].astype("<u4").tobytes())
    return b"".join(out)


def parse_graph_bin(buf):
    """Graph::init (graph/graph.cpp:7-115) -> dict (same keys as preprocess)."""
    o = 0

    def take(fmt):
        nonlocal o
        v = struct.unpack_from(fmt, buf, o)
        o += struct.calcsize(fmt)
        return v

    def arr(dt, n):
        nonlocal o
        a = np.frombuffer(buf, dtype=dt, count=n, offset=o).copy()
        o += a.nbytes
        return a

    N, V, gs, gd = take("<IIII")
    nin, nout, nglob = take("<QQQ")
    g = dict(localVtxCnt=N, globalVtxCnt=V, srcGhostCnt=gs, dstGhostCnt=gd,
             localInEdgeCnt=nin, localOutEdgeCnt=nout, globalEdgeCnt=nglob)
    g["localToGlobal"] = arr("<u4", N)
    g["norm"] = arr("<f4", N)
    sg = arr("<u4", 2 * gs).reshape(-1, 2)
    dg = arr("<u4", 2 * gd).reshape(-1, 2)
    g["srcGhost"], g["srcGhostLocalId"] = sg[:, 0].copy(), sg[:, 1].copy()
    g["dstGhost"], g["dstGhostLocalId"] = dg[:, 0].copy(), dg[:, 1].copy()
    (P,) = take("<I")
    g["numNodes"] = P
    for key in ("fwdLists", "bwdLists"):
        lists = []
        for _ in range(P):
            (c,) = take("<I")
            lists.append(arr("<u4", c))
        g[key] = lists
    cc, nnz = take("<IQ")
    assert cc == N and nnz == nin
    g["cscVal"] = arr("<f4", nnz)
    g["colPtr"] = arr("<u8", N + 1)
    g["rowIdx"] = arr("<u4", nnz)
    rc, nnz = take("<IQ")
    assert rc == N and nnz == nout
    g["csrVal"] = arr("<f4", nnz)
    g["rowPtr"] = arr("<u8", N + 1)
    g["colIdx"] = arr("<u4", nnz)
    assert o == len(buf), (o, len(buf))
    return g
