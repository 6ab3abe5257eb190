"""BASELINE config 0 plumbing: the `graphserver` binary with the reference's command line
(run/run-onnode:154-179) on a Cora-shaped dataset (2708 vertices, 1433-16-7) written in the
reference's file formats: runs on the GPU, writes graph.0.bin byte-identical to the
reference layout, and logs the same validation loss as the oracle epoch."""
import os
import re
import subprocess

import numpy as np
import pytest

import partition_oracle as po

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "dorylus_amd", "graphserver")


def test_graphserver_cora_shaped(tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("graphserver not built")
    import orc
    from helpers import oracle_gcn_epoch
    rng = np.random.default_rng(0)
    V, Eh, dims = 2708, 5278, [1433, 16, 7]
    s, d = rng.integers(0, V, Eh), rng.integers(0, V, Eh)
    src, dst = np.concatenate([s, d]), np.concatenate([d, s])
    X = (rng.random((V, dims[0])) < 0.02).astype(np.float32)          # sparse bag-of-words like Cora
    y = rng.integers(0, dims[-1], V).astype(np.uint32)
    ds = tmp_path / "cora" / "parts_1"
    ds.mkdir(parents=True)
    dsd = str(ds) + "/"
    po.write_bsnap_edges(dsd + "graph.bsnap.edges", V, src, dst)
    po.write_parts(dsd + "graph.bsnap.parts", np.zeros(V, np.int64))
    po.write_features(str(tmp_path / "cora" / "features.bsnap"), X)
    po.write_labels(str(tmp_path / "cora" / "labels.bsnap"), y, dims[-1])
    (tmp_path / "cora.config").write_text("\n".join(map(str, dims)) + "\n")
    cmd = [EXE, "--datasetdir", dsd, "--featuresfile", str(tmp_path / "cora" / "features.bsnap"),
           "--labelsfile", str(tmp_path / "cora" / "labels.bsnap"), "--dshmachinesfile", "/nonexistent",
           "--layerfile", str(tmp_path / "cora.config"), "--pripfile", "/nonexistent", "--dataserverport", "55431",
           "--weightserverport", "65433", "--wserveripfile", "/nonexistent", "--undirected", "0",
           f"--tmpdir={tmp_path}", "--cthreads", "8", "--dthreads", "2", "--dataport", "5000", "--ctrlport", "7000",
           "--nodeport", "6000", "--numlambdas", "1", "--numEpoch", "4", "--validationFrequency", "1", "--MODE", "3",
           "--pipeline", "0", "--staleness", "0", "--gnn", "GCN", "--preprocess", "0", "--timeout_ratio", "1"]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    log = r.stderr
    losses = [float(m.group(2)) for m in re.finditer(r"batch Acc: ([0-9.]+), Loss: ([0-9.]+)", log)]
    assert len(losses) == 4 and losses[-1] < losses[0]
    assert "<EM>: Average  sync epoch time" in log and "<GM>: 2708 global vertices" in log
    # preprocessing output == reference layout (via the reference-pinned oracle)
    g = po.preprocess(src, dst, np.zeros(V, np.int64), 0, 1)
    assert open(dsd + "graph.0.bin", "rb").read() == po.dump_bytes(g)
    assert os.path.exists(dsd + "feats1433.0.bin") and os.path.exists(str(tmp_path / "output_0"))
    # epoch-1 validation loss == oracle epoch with the same Xavier weights
    Ws = [orc.xavier(dims[0], dims[1]), orc.xavier(dims[1], dims[2])]
    T, _ = oracle_gcn_epoch([g], np.zeros(V, np.int64), X, y, Ws, V)
    nval = int(V * 0.1)
    assert abs(losses[0] - T[0]["loss"] / nval) < 1e-4 * max(1.0, T[0]["loss"] / nval)
    # library options without a reference flag pass through as --dory-<option>: the transform-first order gives the
    # same training curve within fp32 rounding, an unknown option is refused
    r2 = subprocess.run(cmd + ["--dory-gcn_transform_first", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r2.returncode == 0, r2.stderr[-2000:]
    losses2 = [float(m.group(2)) for m in re.finditer(r"batch Acc: ([0-9.]+), Loss: ([0-9.]+)", r2.stderr)]
    assert len(losses2) == 4 and np.allclose(losses2, losses, rtol=2e-4)
    r3 = subprocess.run(cmd + ["--dory-no_such_option", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert r3.returncode != 0 and "unknown option" in r3.stderr


def test_run_dorylus_reference_launcher_line(tmp_path):
    """`./run/run-dorylus reddit --l=80 --e=3 gpu --t=0.95 --st=5` (benchmarks/run-reddit-gcn:100) on a toy dataset
    directory under $DORY_FILEPOOL: three epochs run on the device, the ignored knobs are named."""
    from test_graphserver_cli import _toy_filepool, launch
    env = _toy_filepool(tmp_path)
    r = launch(["reddit", "--l=80", "--e=3", "gpu", "--t=0.95", "--st=5"], env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-2000:]
    assert "note: --l=80 --t=0.95 --st=5 ignored by the hip backend" in out
    assert out.count("batch Acc:") == 3 and "<EM>: Average  sync epoch time" in out
