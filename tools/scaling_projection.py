#!/usr/bin/env python3
"""The 2 / 4 / 8-GPU epoch this repo EXPECTS, written down before any multi-GPU node has run it (round 5; projection,
no measurement of the RCCL leg is claimed).  GPU box only:

    python tools/scaling_projection.py [--out profiles/r05_scaling_projection.json] [--cases reddit:uniform reddit:community amazon:uniform]

For every rank r of P in {2, 4, 8} (contiguous-block partitions, exactly bench.py's) it builds rank r's partition on
this one GPU, runs the rank's epoch without the exchange (bench.py --emulate r/P does the same) and records
  * compute ms per epoch, and the part of it that is the local-source launch of each aggregation (the launch an exchange
    in flight hides under: timing family spmm_local_first);
  * nnz, local / ghost vertex counts, and the rows this rank SENDS to every peer per exchange (the reference's
    forwardGhostsList / backwardGhostsList, graph.<id>.bin) -> bytes per peer per exchange = rows x 4 x ld
    (the volume of gcn_ops.cpp:204-282: rows x (4 + 4F) per peer there; no ids travel here, rows are padded to ld).
The model that turns those into a projected epoch (project(), used by bench.py to print multi_gpu.projected beside the
measurement of a real N > 1 run):
  exchange_ms(rank) = max over peers of bytes_to_peer / LINK  (one xGMI link per pair, all pairs at once)
                      + pack and unpack at HBM_EFF + LAT
  exposed_ms        = max(0, exchange_ms - local_first_ms of the aggregation that consumes it)
  epoch_ms(P)       = max over ranks of (compute_ms + sum of exposed_ms over the epoch's exchanges) + allreduce_ms
with LINK = 153 GB/s per direction (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU), HBM_EFF = 5 TB/s (measured, round 6: profiles/r06_pack_rate.json -- pack 5.8-6.1, unpack 4.7-4.8 TB/s of moved bytes on 3.1 M rows of 64 floats; rounds 5's 4 TB/s was a guess) for the
row gathers / scatters of pack / unpack, LAT = 30 us per exchange (launches + group call), and the dW all-reduce as a
ring: 2 (P - 1) / P x bytes / LINK + 2 (P - 1) x 10 us."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LINK_GBPS = 153.0
HBM_EFF_GBPS = 5000.0
LAT_MS = 0.030
AR_HOP_MS = 0.010


def project(ranks, dims, P):
    """ranks: list of per-rank records (see measure_rank).  Returns the projected epoch of the P-way run."""
    L = len(dims) - 1
    ld = [(d + 31) // 32 * 32 for d in dims]
    # exchanges of one GCN epoch (host/engine.cpp runEpoch): forward h_{l-1} before aggregate(l), l = 1..L-1 (width dims[l]);
    # backward grad_l before the backward aggregate of layer l, l = L-1..1 (width dims[l])
    widths = [ld[l] for l in range(1, L)] + [ld[l] for l in range(L - 1, 0, -1)]
    dirs = ["fwd"] * (L - 1) + ["bwd"] * (L - 1)
    per_rank = []
    for r in ranks:
        exposed = 0.0
        ex = []
        for w, d in zip(widths, dirs):
            rows_to = r["send_rows_fwd"] if d == "fwd" else r["send_rows_bwd"]
            rows_in = r["ghosts_src"] if d == "fwd" else r["ghosts_dst"]
            link_ms = max(rows_to) * 4.0 * w / (LINK_GBPS * 1e6) if rows_to else 0.0
            # (a rank also receives: the slowest of its incoming pairs is some peer's outgoing pair, covered by the max over ranks)
            pack_ms = (sum(rows_to) + rows_in) * 2 * 4.0 * w / (HBM_EFF_GBPS * 1e6)
            t = link_ms + pack_ms + LAT_MS
            hide = r["local_first_ms_per_aggregate"]
            e = max(0.0, t - hide)
            exposed += e
            ex.append({"dir": d, "ld": w, "bytes_max_peer": int(max(rows_to) * 4 * w) if rows_to else 0, "exchange_ms": round(t, 4),
                       "hidden_under_ms": round(hide, 4), "exposed_ms": round(e, 4)})
        per_rank.append({"rank": r["rank"], "compute_ms": r["compute_ms"], "exposed_halo_ms": round(exposed, 4),
                         "epoch_ms": round(r["compute_ms"] + exposed, 4), "exchanges": ex})
    wbytes = sum(dims[l] * dims[l + 1] for l in range(L)) * 4
    ar_ms = 2.0 * (P - 1) / P * wbytes / (LINK_GBPS * 1e6) + 2 * (P - 1) * AR_HOP_MS
    worst = max(per_rank, key=lambda x: x["epoch_ms"])
    return {"P": P, "projected_epoch_ms": round(worst["epoch_ms"] + ar_ms, 3), "slowest_rank": worst["rank"],
            "allreduce_ms": round(ar_ms, 4), "compute_ms_max": max(x["compute_ms"] for x in per_rank),
            "compute_ms_mean": round(float(np.mean([x["compute_ms"] for x in per_rank])), 3),
            "exposed_halo_ms_max": max(x["exposed_halo_ms"] for x in per_rank),
            "nnz_in_max_over_mean": round(max(r["nnz_in"] for r in ranks) / max(1.0, float(np.mean([r["nnz_in"] for r in ranks]))), 4),
            "halo_bytes_per_exchange_max_peer": max((e["bytes_max_peer"] for x in per_rank for e in x["exchanges"]), default=0),
            "per_rank": per_rank,
            "model": {"link_GBps": LINK_GBPS, "pack_unpack_GBps": HBM_EFF_GBPS, "exchange_latency_ms": LAT_MS, "allreduce_hop_ms": AR_HOP_MS}}


def measure_rank(da, bench, workload, graph, src, dst, V, dims, r, P, steps, warmup, parts=None):
    if parts is None:
        parts = (np.arange(V, dtype=np.int64) * P // V).astype(np.int32)
    if src is None:   # configs 4 / 5: only the records incident to the rank's block
        s_, d_ = bench.synth_incident_edges(V, bench.WORKLOADS[workload][1], r, P)
        part = da.Partition.build(s_, d_, parts, r, P)
        del s_, d_
    else:
        part = da.Partition.build(src, dst, parts, r, P)
    g = part.view()
    N, Gs, Gd = int(g["localVtxCnt"]), int(g["srcGhostCnt"]), int(g["dstGhostCnt"])
    rec = {"rank": r, "vertices": N, "ghosts_src": Gs, "ghosts_dst": Gd, "nnz_in": int(g["localInEdgeCnt"]),
           "nnz_out": int(g["localOutEdgeCnt"]),
           "send_rows_fwd": [int(len(x)) for x in g["fwdLists"]], "send_rows_bwd": [int(len(x)) for x in g["bwdLists"]]}
    ctx = da.Context(0)
    ctx.configure(da.GCN, dims, V, 0, 1)          # one rank alone: no communicator, ghost rows stay what they are
    ctx.set_option("spmm_blk_force_split", 1)     # the two-launch form of every aggregation, as beside an exchange in flight: its first
    part.upload(ctx, None)                        # launch (local sources: timing family spmm_local_first) is what the exchange hides under
    ctx.preallocate()
    ctx.fill_uniform(0, "x", 1, -1.0, 1.0, g["localToGlobal"])
    if Gs:
        ctx.fill_uniform(0, "fg", 1, -1.0, 1.0, g["srcGhost"])
    labels = np.random.default_rng(2).integers(0, dims[-1], V).astype(np.uint32)
    ctx.labels_upload(labels[g["localToGlobal"]])
    ctx.weights_init_xavier()
    ctx.adam_config(0.01)
    eng = da.NativeEngine(ctx)
    eng.run(warmup)
    ctx.sync()
    ctx.timing_reset()
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    eng.run(steps)
    ctx.sync()
    rec["compute_ms"] = round((time.perf_counter() - t0) * 1e3 / steps, 4)
    ms, n = ctx.timing_get("spmm_local_first")
    rec["local_first_ms_per_aggregate"] = round(ms / n, 4) if n else 0.0
    rec["local_first_launches_per_epoch"] = n / steps
    ms, n = ctx.timing_get("spmm")
    rec["spmm_ms_per_epoch"] = round(ms / steps, 4)
    rec["spmm_gate_timeouts"] = int(ctx.get_option("spmm_gate_timeouts"))
    ctx.timing_enable(False)
    eng.close()
    ctx.close()
    del part
    return rec


def reproject(path):
    """recompute every projection of a stored file from its per-rank measurements with the current model constants (CPU only)"""
    d = json.load(open(path))
    for case in d["cases"].values():
        for P, pr in case["by_P"].items():
            new = project(pr["ranks"], case["dims"], int(P))
            new["ranks"] = pr["ranks"]
            if "single_gpu_epoch_ms" in case:
                new["projected_speedup"] = round(case["single_gpu_epoch_ms"] / new["projected_epoch_ms"], 3)
                new["projected_efficiency"] = round(new["projected_speedup"] / int(P), 3)
            case["by_P"][P] = new
    json.dump(d, open(path, "w"), indent=1)
    return d


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--reproject":
        d = reproject(sys.argv[2])
        print(json.dumps({c: {P: v["projected_epoch_ms"] for P, v in r["by_P"].items()} for c, r in d["cases"].items()}))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_scaling_projection.json"))
    ap.add_argument("--cases", nargs="*", default=["reddit:uniform", "reddit:community", "amazon:uniform"])
    ap.add_argument("--P", type=int, nargs="*", default=[2, 4, 8])
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    import torch  # noqa: F401  (first: one HIP runtime)
    import bench
    import dorylus_amd as da
    out = {"what": __doc__.split("\n\n")[0], "cases": {}}
    for case in a.cases:
        # <workload>:<graph>[:<partitioning>] -- a partitioning names build/parts/<workload>_<graph>_<partitioning>.npy, written by
        # tools/make_partitions.py for the id-SHUFFLED community graph of that size (round 6, review item 3); without it:
        # contiguous blocks, as bench.py partitions
        workload, graph, *pname = case.split(":")
        V, E, dims = bench.WORKLOADS[workload]
        src = dst = None
        parts_vec = None
        if pname:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import make_partitions as mp
            src, dst = mp.community_edges(V, E, 50, 0.85)
            src, dst, _ = mp.shuffled(V, src, dst)
            parts_vec = np.load(os.path.join(ROOT, "build", "parts", f"{workload}_{graph}_{pname[0]}.npy")).astype(np.int32)
        elif workload == "reddit":
            src, dst = bench.synth_edges(graph, V, E)
        res = {"workload": workload, "graph": graph + (" (ids shuffled)" if pname else ""), "partitioning": pname[0] if pname else "contiguous blocks",
               "vertices": V, "dims": dims, "by_P": {}}
        # P = 1 for reference (Reddit only: the whole Amazon graph on one GPU is bench.py --workload amazon)
        if workload == "reddit":
            r1 = measure_rank(da, bench, workload, graph, src, dst, V, dims, 0, 1, a.steps, a.warmup)
            res["single_gpu_epoch_ms"] = r1["compute_ms"]
        for P in a.P:
            ranks = [measure_rank(da, bench, workload, graph, src, dst, V, dims, r, P, a.steps, a.warmup, parts_vec) for r in range(P)]
            pr = project(ranks, dims, P)
            pr["ranks"] = ranks
            if "single_gpu_epoch_ms" in res:
                pr["projected_speedup"] = round(res["single_gpu_epoch_ms"] / pr["projected_epoch_ms"], 3)
                pr["projected_efficiency"] = round(pr["projected_speedup"] / P, 3)
            res["by_P"][str(P)] = pr
            sys.stderr.write(f"{case} P={P}: projected {pr['projected_epoch_ms']} ms (compute max {pr['compute_ms_max']}, exposed halo "
                             f"{pr['exposed_halo_ms_max']}, allreduce {pr['allreduce_ms']})\n")
        del src, dst
        out["cases"][case] = res
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({c: {P: v["projected_epoch_ms"] for P, v in r["by_P"].items()} for c, r in out["cases"].items()}))


if __name__ == "__main__":
    main()
