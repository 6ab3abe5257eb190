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


# ---- run/run-dorylus: the reference's launcher line, verbatim (run/run-dorylus, run/run-onnode:7-20,38-70) ----------
LAUNCH = os.path.join(ROOT, "run", "run-dorylus")


def _toy_filepool(tmp_path, name="reddit"):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import partition_oracle as po
    ds = tmp_path / "filepool" / name
    d = ds / "parts_1"
    d.mkdir(parents=True)
    rng = np.random.default_rng(0)
    V, E, F, C = 60, 300, 6, 3
    s, t = rng.integers(0, V, E).astype(np.uint32), rng.integers(0, V, E).astype(np.uint32)
    po.write_bsnap_edges(str(d / "graph.bsnap.edges"), V, s, t)
    po.write_parts(str(d / "graph.bsnap.parts"), np.zeros(V, np.int64))
    po.write_features(str(ds / "features.bsnap"), rng.random((V, F), dtype=np.float32))
    po.write_labels(str(ds / "labels.bsnap"), rng.integers(0, C, V).astype(np.uint32), C)
    (tmp_path / "layers.config").write_text(f"{F}\n4\n{C}\n")
    env = dict(os.environ, DORY_FILEPOOL=str(tmp_path / "filepool"), DORY_LAYERFILE=str(tmp_path / "layers.config"),
               TMPDIR=str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DORY_NGPUS"):
        env.pop(k, None)
    return env


def launch(args, env):
    return subprocess.run([LAUNCH] + args, capture_output=True, text=True, timeout=300, env=env)


def test_run_dorylus_reference_line(tmp_path):
    """benchmarks/run-reddit-gcn:100 ends in `./run/run-dorylus reddit --l=$NUM_LAMBDA --e=$EPOCHS $STALENESS_FLAGS $MODE
    --t=$TARGET_ACC --st=$SWITCH_THRESHOLD`: the same words reach the graphserver binary; the Lambda / staleness /
    early-stop knobs are ignored with a printed note.  Without a device the binary below fails loudly (no CPU fallback);
    with one the three epochs run (tests/test_gpu_graphserver.py)."""
    import torch
    env = _toy_filepool(tmp_path)
    r = launch(["reddit", "--l=80", "--e=3", "gpu", "--t=0.95", "--st=5"], env)
    out = r.stdout + r.stderr
    assert "note: --l=80 --t=0.95 --st=5 ignored by the hip backend" in out
    if torch.cuda.is_available():
        assert r.returncode == 0, out[-2000:]
        assert out.count("batch Acc:") == 3
    else:
        assert r.returncode != 0
        assert "no HIP device" in out or "no CPU fallback" in out
    # asynchronous flags of the same script ($STALENESS_FLAGS = "--p --s=1"), GAT line (run-reddit-gat:95), cpu token
    r = launch(["reddit", "--l=80", "--e=1", "--p", "--s=1", "cpu", "--t=0.95", "--g=GAT"], env)
    out = r.stdout + r.stderr
    assert "--p --s=1" in out and "unknown argument" not in out


def test_run_dorylus_rejects_typos_and_missing_datasets(tmp_path):
    env = _toy_filepool(tmp_path)
    r = launch(["reddit", "--ee=3", "gpu"], env)
    assert r.returncode == 2 and "unknown argument '--ee=3'" in r.stderr
    r = launch(["reddit", "--e=three"], env)
    assert r.returncode == 2 and "--e wants a number" in r.stderr
    r = launch(["nosuchdata", "--e=3"], env)
    assert r.returncode == 1 and "is empty or not found under" in r.stdout
    r = launch([], env)
    assert r.returncode == 1 and "Usage: run/run-dorylus <Dataset>" in r.stdout
    # the launcher below it: unknown flags are an error too, and the partition directory must exist
    hip = os.path.join(ROOT, "run", "run-dorylus-hip")
    r = subprocess.run([hip, str(tmp_path / "filepool" / "reddit"), "--epochs=3"], capture_output=True, text=True, env=env)
    assert r.returncode == 2 and "unknown argument '--epochs=3'" in r.stderr
    r = subprocess.run([hip, str(tmp_path / "filepool" / "reddit"), "--n=4"], capture_output=True, text=True, env=env)
    assert r.returncode == 1 and "parts_4/ not found" in r.stderr
