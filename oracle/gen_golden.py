#!/usr/bin/env python3
"""gen_golden.py -- TEST INFRASTRUCTURE.  Generates tests/golden/ fixtures.

Runs ONLY in the build container (needs /root/reference).  Two sources:

 1. oracle/_ref/ref_preprocess -- the reference's own DataLoader, compiled
    unmodified from /root/reference (oracle/Makefile) -- produces the
    graph.<id>.bin bytes for toy datasets x P in {1,2,4,8}.  Fixture =
    inputs (graph.bsnap.edges, graph.bsnap.parts) + expected outputs (*.bin).
 2. the reference's Python GCN (miscs/numpy-gnn: load_data.py, layers.py,
    loss.py) is IMPORTED from /root/reference and run on a toy symmetric
    graph; fixture = inputs + every intermediate tensor of one 2-layer GCN
    forward/backward (float64 results stored as float64).

Fixtures are data only (inputs and expected outputs); no reference source is
copied.  Re-run:  python oracle/gen_golden.py
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
sys.path.insert(0, HERE)
import partition_oracle as po  # noqa: E402


def toy_edges(seed, V, E, hub=False, symmetric=False, dedup=False):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, V, E)
    dst = rng.integers(0, V, E)
    if hub == "big":  # > 8192 edges on one destination and on one source
        dst[:11000] = 7
        src[11000:21000] = 2900
    elif hub:  # one high in-degree vertex + one high out-degree vertex
        k = E // 4
        dst[:k] = 3 % V
        src[k:2 * k] = 5 % V
    if symmetric:
        keep = src != dst
        src, dst = src[keep], dst[keep]
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    if dedup:
        key = np.unique(src.astype(np.int64) * V + dst)
        src, dst = key // V, key % V
    return src.astype(np.uint32), dst.astype(np.uint32)


def gen_partition_fixtures():
    cases = [
        # name, V, E, P, undirected, hub, partition kind
        ("toy60_p1", 60, 500, 1, 0, True, "block"),
        ("toy60_p2", 60, 500, 2, 0, True, "block"),
        ("toy60_p4_hash", 60, 500, 4, 0, True, "hash"),
        ("toy97_p8_und", 97, 700, 8, 1, False, "rand"),
        ("toy40_p3_empty", 40, 200, 3, 0, False, "skip1"),  # partition 1 owns nothing
        # one destination with > 8192 in-edges and one source with > 8192 out-edges (K1's long-row clamp and the
        # blocked kernels' hub segments run on bytes the reference's DataLoader wrote), two partitions with ghosts
        ("hub3000_p2", 3000, 52000, 2, 0, "big", "block"),
    ]
    exe = os.path.join(HERE, "_ref", "ref_preprocess")
    for name, V, E, P, und, hub, kind in cases:
        d = os.path.join(GOLD, "parts_" + name)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        src, dst = toy_edges(len(name) * 7 + V, V, E, hub=hub)
        rng = np.random.default_rng(V + P)
        if kind == "block":
            parts = (np.arange(V) * P) // V
        elif kind == "hash":
            parts = (np.arange(V) * 2654435761 % 4294967296) % P
        elif kind == "rand":
            parts = rng.integers(0, P, V)
        else:  # skip1
            parts = np.where(np.arange(V) % 2 == 0, 0, 2)
        po.write_bsnap_edges(os.path.join(d, "graph.bsnap.edges"), V, src, dst)
        po.write_parts(os.path.join(d, "graph.bsnap.parts"), parts)
        with open(os.path.join(d, "meta.txt"), "w") as f:
            f.write(f"V={V} E={E} P={P} undirected={und}\n")
        for nid in range(P):
            subprocess.run([exe, d + "/", str(nid), str(P), str(und)], check=True,
                           capture_output=True)
        print("partition fixture", name, sorted(os.listdir(d)))


def gen_numpy_gnn_fixture():
    """One epoch of the reference's numpy GCN (miscs/numpy-gnn) on a toy graph."""
    sys.path.insert(0, os.path.join(REF, "miscs", "numpy-gnn"))
    import layers as ref_layers  # reference code, imported in place
    import load_data as ref_load
    import loss as ref_loss

    V, F0, F1, C = 48, 10, 6, 4
    # numpy-gnn builds A_hat from a 0/1 dense matrix with out-degree norm
    # (load_data.py:36-43), so it coincides with the C++ in-degree / duplicate-
    # keeping definition only on symmetric, duplicate-free graphs: use one.
    src, dst = toy_edges(11, V, 160, symmetric=True, dedup=True)
    rng = np.random.default_rng(5)
    X = rng.uniform(-1, 1, (V, F0)).astype(np.float32)
    labels = rng.integers(0, C, V).astype(np.uint32)
    W0 = (rng.standard_normal((F0, F1)) / np.sqrt(F0)).astype(np.float32)
    W1 = (rng.standard_normal((F1, C)) / np.sqrt(F1)).astype(np.float32)

    with tempfile.TemporaryDirectory() as d:
        d += "/"
        po.write_bsnap_edges(d + "graph.bsnap", V, src, dst)
        po.write_features(d + "features.bsnap", X)
        po.write_labels(d + "labels.bsnap", labels, C)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            A_hat, feats, tl = ref_load.load_data(d, "toy", binary=True)
    onehot = np.eye(C)[tl]

    agg0 = ref_layers.Aggregate("A0", A_hat)
    lin0 = ref_layers.Linear("W0", F0, F1, "xavier").set_W(W0.astype(np.float64))
    act0 = ref_layers.Tanh("T0")
    agg1 = ref_layers.Aggregate("A1", A_hat)
    lin1 = ref_layers.Linear("W1", F1, C, "xavier").set_W(W1.astype(np.float64))
    lossf = ref_loss.SoftmaxCrossEntropyLoss("loss")

    ah0 = agg0.forward(feats.astype(np.float64))
    z0 = lin0.forward(ah0)
    h0 = act0.forward(z0)
    ah1 = agg1.forward(h0)
    z1 = lin1.forward(ah1)
    d = lossf.backward(z1, onehot)            # softmax(z1) - onehot
    grad1 = lin1.backward(d)                  # d @ W1.T ; lin1.grad_W = ah1.T @ d
    aTg0 = agg1.backward(grad1)               # A_hat.T @ grad1
    g0 = act0.backward(aTg0)                  # aTg0 * (1 - h0^2)
    lin0.backward(g0)                         # lin0.grad_W = ah0.T @ g0

    np.savez_compressed(
        os.path.join(GOLD, "numpy_gnn_epoch.npz"),
        V=V, src=src, dst=dst, X=X, labels=labels, W0=W0, W1=W1, A_hat=A_hat,
        ah0=ah0, z0=z0, h0=h0, ah1=ah1, z1=z1, d=d, grad1=grad1, dW1=lin1.grad_W,
        aTg0=aTg0, g0=g0, dW0=lin0.grad_W)
    print("numpy-gnn fixture written")


def gen_numpy_gnn_large(name, dims, V=1500, E=18000, seed=23, rows=96, hub=False):
    """The reference's numpy GCN at ~2 000 vertices with the layer widths of a BASELINE config (any depth).  Kept small:
    inputs are regenerated from the stored seed-free arrays (edges, X, labels, W), expected outputs are float32 rows of
    a fixed sample of vertices for every intermediate tensor plus the complete weight gradients."""
    sys.path.insert(0, os.path.join(REF, "miscs", "numpy-gnn"))
    import layers as ref_layers
    import load_data as ref_load
    import loss as ref_loss
    L = len(dims) - 1
    src, dst = toy_edges(seed, V, E, hub=hub, symmetric=True, dedup=True)   # hub: vertices 3 and 5 neighbour most of the graph
    rng = np.random.default_rng(seed + 1)
    Xq = rng.integers(-64, 65, (V, dims[0])).astype(np.int8)      # features in 1/64 steps: one byte each in the fixture
    X = Xq.astype(np.float32) / np.float32(64)
    labels = rng.integers(0, dims[-1], V).astype(np.uint32)
    Ws = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(L)]
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        po.write_bsnap_edges(d + "graph.bsnap", V, src, dst)
        po.write_features(d + "features.bsnap", X)
        po.write_labels(d + "labels.bsnap", labels, dims[-1])
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            A_hat, feats, tl = ref_load.load_data(d, "toy", binary=True)
    onehot = np.eye(dims[-1])[tl]
    aggs = [ref_layers.Aggregate(f"A{l}", A_hat) for l in range(L)]
    lins = [ref_layers.Linear(f"W{l}", dims[l], dims[l + 1], "xavier").set_W(Ws[l].astype(np.float64)) for l in range(L)]
    acts = [ref_layers.Tanh(f"T{l}") for l in range(L - 1)]
    lossf = ref_loss.SoftmaxCrossEntropyLoss("loss")
    out = {}
    h = feats.astype(np.float64)
    for l in range(L):
        out[f"ah{l}"] = aggs[l].forward(h)
        out[f"z{l}"] = lins[l].forward(out[f"ah{l}"])
        if l < L - 1:
            h = out[f"h{l}"] = acts[l].forward(out[f"z{l}"])
    out["d"] = lossf.backward(out[f"z{L-1}"], onehot)
    grad = lins[L - 1].backward(out["d"])
    out[f"dW{L-1}"] = lins[L - 1].grad_W
    for l in range(L - 1, 0, -1):
        out[f"grad{l}"] = grad
        out[f"aTg{l-1}"] = aggs[l].backward(grad)
        out[f"g{l-1}"] = acts[l - 1].backward(out[f"aTg{l-1}"])
        grad = lins[l - 1].backward(out[f"g{l-1}"])
        out[f"dW{l-1}"] = lins[l - 1].grad_W
    sample = np.sort(np.random.default_rng(seed + 2).choice(V, rows, replace=False))
    if hub:
        sample = np.unique(np.concatenate([sample, [3, 5]]))          # the hub rows are always among the compared rows
    et = np.uint16 if V <= 65536 and V > 8192 else np.uint32       # (the 20 000-vertex fixture: two bytes per endpoint)
    store = {"V": V, "src": src.astype(et), "dst": dst.astype(et), "X_q64": Xq, "labels": labels.astype(np.uint8 if dims[-1] < 256 and V > 8192 else np.uint32),
             "sample": sample, "dims": np.asarray(dims)}
    for l in range(L):
        store[f"W{l}"] = Ws[l]
    for k, v in out.items():
        store[k] = v.astype(np.float32) if k.startswith("dW") else v[sample].astype(np.float32)
    # the complete gradient entering the backward half (the GPU test uploads it as "grad"@(L-1) and compares aTg, g, grad
    # and dW of the layers below with the rows above directly)
    store[f"grad{L-1}_full"] = out[f"grad{L-1}"].astype(np.float32)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **store)
    print("numpy-gnn fixture", name, "written:", os.path.getsize(os.path.join(GOLD, name + ".npz")) // 1024, "KiB")


def gen_wire_fixture():
    """message headers of the weight-server / Lambda protocol, bytes written by the reference's own serialisation
    code (oracle/ref_wire.cpp includes /root/reference/src/common/utils.hpp)"""
    exe = os.path.join(HERE, "_ref", "ref_wire")
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    import json
    json.loads(out)
    with open(os.path.join(GOLD, "wire_headers.json"), "w") as f:
        f.write(out)
    print("wire fixture written")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    gen_wire_fixture()
    gen_partition_fixtures()
    gen_numpy_gnn_fixture()
    gen_numpy_gnn_large("numpy_gnn_reddit_dims", [602, 128, 41])            # BASELINE config 2 widths
    gen_numpy_gnn_large("numpy_gnn_amazon_dims", [300, 64, 64, 25], seed=29)  # config 4: 3 layers
    # round 5: 4 096 vertices (the dense A_hat of the Python model still fits) with two hub rows of ~3 500 neighbours
    gen_numpy_gnn_large("numpy_gnn_hub4k", [602, 128, 41], V=4096, E=40000, seed=31, rows=128, hub=True)
    # round 6: 20 000 vertices, ~590 000 edges, hub rows of ~13 000 neighbours (beyond K1's long-row clamp of 8 192) -- the largest
    # graph the Python model's dense A_hat (3.2 GB) handles in minutes; widths 128-64-16
    gen_numpy_gnn_large("numpy_gnn_20k", [128, 64, 16], V=20000, E=300000, seed=37, rows=192, hub=True)
