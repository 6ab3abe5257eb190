"""Host logic (CPU): the C++ partition builder (host/partition.cpp) is index-identical
to the reference's DataLoader -- byte-for-byte equal graph.<id>.bin files (golden
fixtures written by oracle/_ref/ref_preprocess) -- and round-trips through its reader."""
import glob
import os
import subprocess
import tempfile

import numpy as np
import pytest

import partition_oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "parts_*")))


@pytest.fixture(scope="module")
def da():
    import dorylus_amd
    if not os.path.exists(dorylus_amd.LIB_PATH):
        pytest.skip("library not built")
    return dorylus_amd


def _meta(d):
    kv = dict(t.split("=") for t in open(os.path.join(d, "meta.txt")).read().split())
    return int(kv["P"]), bool(int(kv["undirected"]))


@pytest.mark.parametrize("d", CASES, ids=[os.path.basename(c) for c in CASES])
def test_builder_bytes_equal_reference(da, d, tmp_path):
    P, und = _meta(d)
    for nid in range(P):
        ref = open(os.path.join(d, f"graph.{nid}.bin"), "rb").read()
        part = da.Partition.build_from_files(d + "/", nid, P, und)
        out = str(tmp_path / f"graph.{nid}.bin")
        part.save(out)
        assert open(out, "rb").read() == ref
        # reader (Graph::init) round trip: load the reference's file, save again
        again = da.Partition.load(os.path.join(d, f"graph.{nid}.bin"))
        again.save(out)
        assert open(out, "rb").read() == ref
        v, g = part.view(), po.parse_graph_bin(ref)
        for k in ("colPtr", "rowIdx", "cscVal", "rowPtr", "colIdx", "csrVal", "norm", "srcGhost", "dstGhost"):
            assert np.array_equal(v[k], g[k]), k


@pytest.mark.parametrize("seed", range(6))
def test_builder_matches_oracle_random(da, seed):
    """in-memory entry point vs the (reference-pinned) numpy oracle, ragged cases included"""
    rng = np.random.default_rng(100 + seed)
    V = int(rng.integers(1, 400))
    E = int(rng.integers(0, 6000))
    P = int(rng.integers(1, 9))
    und = bool(seed % 2)
    src, dst = rng.integers(0, V, E), rng.integers(0, V, E)
    parts = rng.integers(0, P, V)
    if seed == 5:
        parts[:] = 0          # every other partition is empty
    for nid in range(P):
        part = da.Partition.build(src, dst, parts, nid, P, und)
        with tempfile.NamedTemporaryFile() as f:
            part.save(f.name)
            assert open(f.name, "rb").read() == po.dump_bytes(po.preprocess(src, dst, parts, nid, P, und))


def test_builder_rejects_bad_input(da):
    with pytest.raises(da.DoryError):
        da.Partition.build([0, 5], [1, 2], [0, 0, 0], 0, 1)       # vertex id 5 >= V
    with pytest.raises(da.DoryError):
        da.Partition.build([0], [1], [0, 3], 0, 2)                # partition id 3 >= P
    with pytest.raises(da.DoryError):
        da.Partition.load("/nonexistent/graph.0.bin")


@pytest.mark.parametrize("threads", [2, 5, 16])
def test_builder_parallel_passes_are_order_preserving(da, threads, monkeypatch):
    """the vertex-ownership parallel passes give the same bytes as the sequential build / the oracle"""
    monkeypatch.setenv("DORY_BUILD_THREADS", str(threads))
    rng = np.random.default_rng(threads)
    V, E, P = 333, 9000, 3
    src, dst = rng.integers(0, V, E), rng.integers(0, V, E)
    src[:500] = 7                                   # hub with many duplicate edges: order matters
    parts = rng.integers(0, P, V)
    for und in (False, True):
        for nid in range(P):
            part = da.Partition.build(src, dst, parts, nid, P, und)
            with tempfile.NamedTemporaryFile() as f:
                part.save(f.name)
                assert open(f.name, "rb").read() == po.dump_bytes(po.preprocess(src, dst, parts, nid, P, und))
