"""Host logic (CPU): readers of the reference's input files (layer config, features.bsnap +
feats cache, labels.bsnap) against data written in the documented formats
and against the layer files the reference ships (fixture copies of their *contents* are
not needed: the formats are one integer per line)."""
import os

import numpy as np
import pytest

import partition_oracle as po


@pytest.fixture(scope="module")
def da():
    import dorylus_amd
    if not os.path.exists(dorylus_amd.LIB_PATH):
        pytest.skip("library not built")
    return dorylus_amd


def test_layer_config(da, tmp_path):
    p = tmp_path / "reddit.config"
    p.write_text("602\n128\n41\n")
    assert da.read_layer_config(str(p)) == [602, 128, 41]
    p.write_text("  1433 \n\n16\r\n7\n\n")
    assert da.read_layer_config(str(p)) == [1433, 16, 7]
    p.write_text("5\n")
    with pytest.raises(da.DoryError):
        da.read_layer_config(str(p))
    with pytest.raises(da.DoryError):
        da.read_layer_config(str(tmp_path / "missing.config"))


@pytest.mark.parametrize("P", [1, 3])
def test_features_and_labels(da, tmp_path, P):
    rng = np.random.default_rng(P)
    V, E, F, K = 90, 500, 11, 5
    src, dst = rng.integers(0, V, E), rng.integers(0, V, E)
    parts = rng.integers(0, P, V)
    X = rng.standard_normal((V, F)).astype(np.float32)
    y = rng.integers(0, K, V).astype(np.uint32)
    d = str(tmp_path) + "/"
    po.write_features(d + "features.bsnap", X)
    po.write_labels(d + "labels.bsnap", y, K)
    for nid in range(P):
        part = da.Partition.build(src, dst, parts, nid, P)
        g = part.view()
        local, ghost = da.read_features(d + "features.bsnap", part, F, nid, cache_dir=d)
        assert np.array_equal(local, X[g["localToGlobal"]])
        assert np.array_equal(ghost, X[g["srcGhost"]].reshape(len(g["srcGhost"]), F))
        # cache file = local block then ghost block (engine/utils.cpp:548-549), and is used next time
        cache = d + f"feats{F}.{nid}.bin"
        raw = np.fromfile(cache, np.float32)
        assert np.array_equal(raw, np.concatenate([local.ravel(), ghost.ravel()]))
        os.rename(d + "features.bsnap", d + "features.moved")
        l2, g2 = da.read_features(d + "features.bsnap", part, F, nid, cache_dir=d)
        assert np.array_equal(l2, local) and np.array_equal(g2, ghost)
        os.rename(d + "features.moved", d + "features.bsnap")
        assert np.array_equal(da.read_labels(d + "labels.bsnap", part, K), y[g["localToGlobal"]])
        with pytest.raises(da.DoryError):
            da.read_features(d + "features.bsnap", part, F + 1, nid)      # header mismatch
        with pytest.raises(da.DoryError):
            da.read_labels(d + "labels.bsnap", part, K + 1)
