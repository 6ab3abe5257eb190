"""Oracle pin #1: the numpy restatement of DataLoader::preprocess reproduces, byte
for byte, the graph.<id>.bin files written by the reference's own DataLoader
(oracle/_ref/ref_preprocess; fixtures committed under tests/golden/parts_*)."""
import glob
import os
import subprocess
import tempfile

import numpy as np
import pytest

import partition_oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "parts_*")))


def _meta(d):
    kv = dict(t.split("=") for t in open(os.path.join(d, "meta.txt")).read().split())
    return int(kv["P"]), bool(int(kv["undirected"]))


@pytest.mark.parametrize("d", CASES, ids=[os.path.basename(c) for c in CASES])
def test_oracle_matches_reference_bins(d):
    P, und = _meta(d)
    V, src, dst = po.read_bsnap_edges(os.path.join(d, "graph.bsnap.edges"))
    parts = np.loadtxt(os.path.join(d, "graph.bsnap.parts"), dtype=np.int64, ndmin=1)
    assert parts.size == V
    for nid in range(P):
        ref = open(os.path.join(d, f"graph.{nid}.bin"), "rb").read()
        g = po.preprocess(src, dst, parts, nid, P, und)
        assert po.dump_bytes(g) == ref
        # reader round trip (Graph::init, graph.cpp:7-115)
        back = po.parse_graph_bin(ref)
        assert np.array_equal(back["rowIdx"], g["rowIdx"])
        assert np.array_equal(back["srcGhostLocalId"], g["localVtxCnt"] + np.arange(g["srcGhostCnt"]))


REF_EXE = os.path.join(ROOT, "oracle", "_ref", "ref_preprocess")


@pytest.mark.skipif(not os.path.exists(REF_EXE), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed", range(4))
def test_oracle_matches_live_reference(seed):
    """Random graphs through the live reference binary (when present)."""
    rng = np.random.default_rng(seed)
    V = int(rng.integers(5, 300))
    E = int(rng.integers(0, 4000))
    P = int(rng.integers(1, 6))
    und = bool(seed % 2)
    src, dst = rng.integers(0, V, E), rng.integers(0, V, E)
    parts = rng.integers(0, P, V)
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        po.write_bsnap_edges(d + "graph.bsnap.edges", V, src, dst)
        po.write_parts(d + "graph.bsnap.parts", parts)
        for nid in range(P):
            subprocess.run([REF_EXE, d, str(nid), str(P), str(int(und))], check=True, capture_output=True)
            ref = open(d + f"graph.{nid}.bin", "rb").read()
            assert po.dump_bytes(po.preprocess(src, dst, parts, nid, P, und)) == ref
