"""Parity at BASELINE.json's full size (Reddit: 232 965 vertices, ~114.6 M edges, F = 602/128)
through size-independent properties + sampled rows against the oracle:
  * checksum of checksums: 1^T (A_hat X) == (1^T A_hat) X, right side in float64 on the host;
  * linearity / scale covariance: aggregate(2x) == 2 * aggregate(x) bit-exactly (power of two);
  * K1 (edge order) vs K1b (source-blocked) agree to fp32 reassociation;
  * 2 000 sampled destination rows vs the CPU oracle, forward (CSC) and backward (CSR);
  * a full epoch keeps finite values, and the partition is index-consistent (CSR == CSC^T)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def reddit():
    import dorylus_amd as da
    from bench import REDDIT_E, REDDIT_V, synth_edges
    src, dst = synth_edges("uniform", REDDIT_V, REDDIT_E)
    part = da.Partition.build(src, dst, np.zeros(REDDIT_V, np.int32), 0, 1)
    del src, dst
    return da, part, part.view()


def _oracle_rows(g, ptr_key, idx_key, val_key, X, rows):
    """oracle aggregate for a sample of rows (each with its complete edge list)"""
    import orc
    ptr = g[ptr_key]
    segs = [(int(ptr[v]), int(ptr[v + 1])) for v in rows]
    sub_ptr = np.concatenate([[0], np.cumsum([b - a for a, b in segs])]).astype(np.uint64)
    idx = np.concatenate([g[idx_key][a:b] for a, b in segs]).astype(np.uint32)
    val = np.concatenate([g[val_key][a:b] for a, b in segs]).astype(np.float32)
    n = len(rows)
    F = X.shape[1]
    # the oracle treats indices >= n as "ghost" rows: put X behind the sampled rows
    out = np.empty((n, F), np.float32)
    orc.lib.orc_aggregate_gcn(n, F, sub_ptr, idx + np.uint32(n), val, np.ascontiguousarray(g["norm"][rows]),
                              np.ascontiguousarray(X[rows]), X, out)
    return out


@pytest.mark.parametrize("F", [602, 128])
def test_fullscale_aggregate_properties(reddit, F):
    da, part, g = reddit
    from helpers import rel_err
    N = int(g["localVtxCnt"])
    ctx = da.Context(0)
    ctx.configure(da.GCN, [F, F, 3], N)
    part.upload(ctx)
    ctx.preallocate()
    ctx.fill_uniform(0, "x", 11)
    ctx.fill_uniform(1, "grad", 12)
    X = ctx.download(0, "x")
    G = ctx.download(1, "grad")
    outs = {}
    for variant in (0, 1):
        ctx.set_option("spmm_variant", variant)
        ctx.aggregate(0, da.FORWARD)
        ctx.aggregate(1, da.BACKWARD)
        outs[variant] = (ctx.download(0, "ah"), ctx.download(0, "aTg"))
    # K1 vs K1b
    assert rel_err(outs[1][0], outs[0][0]) < 1e-5 and rel_err(outs[1][1], outs[0][1]) < 1e-5
    # sampled rows vs oracle (highest-degree rows included)
    rng = np.random.default_rng(0)
    deg = np.diff(g["colPtr"].astype(np.int64))
    rows = np.unique(np.concatenate([rng.integers(0, N, 2000), np.argsort(deg)[-20:], [0, N - 1]]))
    ref = _oracle_rows(g, "colPtr", "rowIdx", "cscVal", X, rows)
    for variant in (0, 1):
        assert rel_err(outs[variant][0][rows], ref) < 1e-4
    refb = _oracle_rows(g, "rowPtr", "colIdx", "csrVal", G, rows)
    for variant in (0, 1):
        assert rel_err(outs[variant][1][rows], refb) < 1e-4
    # checksum of checksums in float64
    w = g["norm"].astype(np.float64) + np.bincount(g["rowIdx"], weights=g["cscVal"].astype(np.float64), minlength=N)
    expect = w @ X.astype(np.float64)
    for variant in (0, 1):
        got = outs[variant][0].astype(np.float64).sum(0)
        assert np.abs(got - expect).max() / np.abs(expect).max() < 1e-5
    # scale covariance, bit-exact
    ctx.upload(0, "x", X * np.float32(2.0))
    for variant in (0, 1):
        ctx.set_option("spmm_variant", variant)
        ctx.aggregate(0, da.FORWARD)
        assert np.array_equal(ctx.download(0, "ah"), outs[variant][0] * np.float32(2.0))
    ctx.close()


def test_fullscale_partition_consistency(reddit):
    """CSR is the transpose of CSC: same multiset of (src, dst, value) -- index-exact."""
    _, _, g = reddit
    N = int(g["localVtxCnt"])
    dst_c = np.repeat(np.arange(N, dtype=np.int64), np.diff(g["colPtr"].astype(np.int64)))
    src_r = np.repeat(np.arange(N, dtype=np.int64), np.diff(g["rowPtr"].astype(np.int64)))
    kc = g["rowIdx"].astype(np.int64) * N + dst_c
    kr = src_r * N + g["colIdx"].astype(np.int64)
    oc, orr = np.argsort(kc, kind="stable"), np.argsort(kr, kind="stable")
    assert np.array_equal(kc[oc], kr[orr])
    assert np.array_equal(g["cscVal"][oc], g["csrVal"][orr])
    assert int(g["localInEdgeCnt"]) == int(g["localOutEdgeCnt"]) == int(g["globalEdgeCnt"])


def test_fullscale_epoch_finite_and_learning(reddit):
    da, part, g = reddit
    N = int(g["localVtxCnt"])
    ctx = da.Context(0)
    ctx.configure(da.GCN, [602, 128, 41], N)
    part.upload(ctx)
    ctx.preallocate()
    ctx.fill_uniform(0, "x", 1)
    ctx.labels_upload(np.random.default_rng(2).integers(0, 41, N).astype(np.uint32))
    ctx.weights_init_xavier()
    ctx.adam_config(0.01)
    eng = da.NativeEngine(ctx)
    losses = []
    for _ in range(4):
        eng.run(1)
        a, l, n = ctx.train_stat()
        assert np.isfinite(l) and n == int(N * 0.1)
        losses.append(l / n)
    assert losses[-1] < losses[0]          # Adam on the summed gradients reduces the validation loss
    assert np.isfinite(ctx.weight_get(0)).all() and np.isfinite(ctx.download(0, "aTg")).all()
    eng.close()
    ctx.close()
