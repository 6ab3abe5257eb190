#!/usr/bin/env python3
"""What do pack and unpack (K6: gather the rows a rank sends / scatter the rows it receives) really run at?  tools/scaling_projection.py
prices them at 4 TB/s of moved bytes; this measures them on one rank of 8 of the Amazon-size community graph under the ldg
partition (3.1 M rows each way, 64 floats).  GPU box only:  python tools/pack_rate.py [--parts ldg10]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", default="ldg10")
    ap.add_argument("--rank", type=int, default=0)
    a = ap.parse_args()
    import torch
    import bench
    import dorylus_amd as da
    import make_partitions as mp
    V, E, dims = bench.WORKLOADS["amazon"]
    src, dst = mp.community_edges(V, E, 50, 0.85)
    src, dst, _ = mp.shuffled(V, src, dst)
    parts = np.load(os.path.join(ROOT, "build", "parts", f"amazon_community_{a.parts}.npy")).astype(np.int32)
    part = da.Partition.build(src, dst, parts, a.rank, 8)
    del src, dst
    g = part.view()
    ctx = da.Context(0)
    ctx.configure(da.GCN, dims, V, a.rank, 8)
    part.upload(ctx, parts)
    ctx.preallocate()
    ctx.fill_uniform(0, "x", 1, -1.0, 1.0, g["localToGlobal"])
    ctx.aggregate(0, da.FORWARD)
    ctx.apply_vertex(0, da.FORWARD)          # h@0 exists
    out = {"partitioning": a.parts, "rank": a.rank}
    for layer, dd, what in ((1, da.FORWARD, "forward (h@0 -> fg@1)"), (1, da.BACKWARD, "backward (grad@1 -> bg@0)")):
        send_rows = int(sum(len(x) for x in (g["fwdLists"] if dd == da.FORWARD else g["bwdLists"])))
        recv_rows = int(g["srcGhostCnt"] if dd == da.FORWARD else g["dstGhostCnt"])
        ld = 64
        sbuf = torch.empty(max(1, send_rows) * ld, device="cuda")
        rbuf = torch.zeros(max(1, recv_rows) * ld, device="cuda")
        torch.cuda.synchronize()
        for fn, buf, rows, name in ((ctx.halo_pack, sbuf, send_rows, "pack"), (ctx.halo_unpack, rbuf, recv_rows, "unpack")):
            fn(layer, dd, buf.data_ptr())
            ctx.sync()
            ctx.timing_reset()
            ctx.timing_enable(True)
            for _ in range(10):
                fn(layer, dd, buf.data_ptr())
            ctx.sync()
            ms, n = ctx.timing_get("halo")
            ctx.timing_enable(False)
            t = ms / n * 1e-3
            out[f"{name} {what}"] = {"rows": rows, "ms": round(t * 1e3, 4), "moved_GBps": round(2 * rows * ld * 4 / t / 1e9, 1)}
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
