"""graphserver binary (host/graphserver_main.cpp, the reference's command line: run/run-onnode:154-179) without a
GPU: argument handling and the loud failure when no device is there (the product path has no CPU fallback)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "dorylus_amd", "graphserver")


def run(args, **kw):
    return subprocess.run([BIN] + args, capture_output=True, text=True, timeout=120, **kw)


def test_usage_and_missing_files():
    r = run([])
    assert "usage: graphserver --datasetdir" in (r.stdout + r.stderr)
    r = run(["--datasetdir", "/nonexistent/", "--featuresfile", "x", "--labelsfile", "y", "--layerfile", "z",
             "--numEpochs", "1", "--gnn", "GCN", "--tmpdir", "/tmp"])
    assert r.returncode != 0
    assert "cannot open layer configuration file z" in (r.stdout + r.stderr)


def test_fails_loudly_without_a_device(tmp_path):
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import partition_oracle as po
    d = tmp_path / "parts_1"
    d.mkdir()
    rng = np.random.default_rng(0)
    V, E, F, C = 50, 200, 6, 3
    s, t = rng.integers(0, V, E).astype(np.uint32), rng.integers(0, V, E).astype(np.uint32)
    po.write_bsnap_edges(str(d / "graph.bsnap.edges"), V, s, t)
    po.write_parts(str(d / "graph.bsnap.parts"), np.zeros(V, np.int64))
    po.write_features(str(tmp_path / "features.bsnap"), rng.random((V, F), dtype=np.float32))
    po.write_labels(str(tmp_path / "labels.bsnap"), rng.integers(0, C, V).astype(np.uint32), C)
    (tmp_path / "layers.config").write_text(f"{F}\n4\n{C}\n")
    r = run(["--datasetdir", str(d) + "/", "--featuresfile", str(tmp_path / "features.bsnap"),
             "--labelsfile", str(tmp_path / "labels.bsnap"), "--layerfile", str(tmp_path / "layers.config"),
             "--numEpochs", "1", "--gnn", "GCN", "--tmpdir", str(tmp_path)])
    assert r.returncode != 0
    assert "no HIP device" in (r.stdout + r.stderr) or "no CPU fallback" in (r.stdout + r.stderr)
