#!/usr/bin/env python3
"""Launch-bound regime: epoch time of a Cora-shaped synthetic graph (2 708 vertices, 10 556
edges, 1433-16-7; BASELINE.json configs[0] shape) eager vs. replayed epoch graph.  GPU box only.
  python tools/bench_small.py [--gnn gcn|gat|gatmh] [--epochs 200]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (first: one HIP runtime)
import dorylus_amd as da  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gnn", default="gcn", choices=["gcn", "gat", "gatmh"])
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--verts", type=int, default=2708)
    ap.add_argument("--edges", type=int, default=5278, help="undirected edges (doubled)")
    ap.add_argument("--dims", type=int, nargs="+", default=[1433, 16, 7])
    a = ap.parse_args()
    rng = np.random.default_rng(5)
    s, d = rng.integers(0, a.verts, a.edges), rng.integers(0, a.verts, a.edges)
    keep = s != d
    s, d = np.concatenate([s[keep], d[keep]]).astype(np.uint32), np.concatenate([d[keep], s[keep]]).astype(np.uint32)
    part = da.Partition.build(s, d, np.zeros(a.verts, np.int32), 0, 1)
    g = part.view()
    gnn = {"gcn": da.GCN, "gat": da.GAT, "gatmh": da.GATMH}[a.gnn]
    for graph in (0, 1):
        ctx = da.Context(0)
        ctx.configure(gnn, a.dims, a.verts)
        if a.gnn == "gatmh":
            ctx.gatmh_heads([4, 1])
        part.upload(ctx)
        ctx.preallocate()
        ctx.fill_uniform(0, "x" if a.gnn == "gcn" else "h", 3, -1.0, 1.0, g["localToGlobal"])
        ctx.labels_upload(rng.integers(0, a.dims[-1], a.verts).astype(np.uint32))
        ctx.weights_init_xavier()
        ctx.adam_config(0.01)
        ctx.set_option("epoch_graph", graph)
        eng = da.NativeEngine(ctx)
        eng.run(5)
        ms = eng.run(a.epochs)
        print(f"{a.gnn} {a.verts} verts {len(s)} edges dims {a.dims} epoch_graph={graph}: "
              f"median {np.median(ms)*1e3:.1f} us/epoch, min {ms.min()*1e3:.1f}", flush=True)
        eng.close()
        ctx.close()


if __name__ == "__main__":
    main()
