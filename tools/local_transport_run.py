#!/usr/bin/env python3
"""The overlapped halo schedule with real concurrency on ONE GPU (round 6, review item 2): P ranks of one process over the
in-process device transport (dory_comm_init_local), one host thread per rank, whole epochs inside the C++ Engine.

    python tools/local_transport_run.py [--V 200000 --E 4000000 --dims 128 128 16 --epochs 50] > profiles/r06_local_transport.json

For a random graph in contiguous blocks over P = 2 and 4 ranks, halo_overlap 0 and 1, 50 epochs back to back: epoch time,
the comm-stream intervals (halo = pack + device-to-device copies + unpack; halo_deferred = the part an aggregation is allowed
to run beside), the local-source launch that runs beside them (spmm_beside_halo), their intersection on the device's clock
(halo_hidden), K1s's gate counters while copies run beside it, and whether overlap on / off end in the same bits.  The
ranks SHARE the device: each rank's sweeps take spmm_sweep_cus = 32 / P - 2 CUs of every XCD.  A second configuration gives
rank 0 98 % of the vertices: rank 0's K1s beside its own comm stream
with next to nothing else on the device -- the closest one GPU gets to one rank of a real run (24 CUs of every XCD to rank 0's
sweeps, 4 to rank 1's, 4 left to the copies)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=200000)
    ap.add_argument("--E", type=int, default=4000000)
    ap.add_argument("--dims", type=int, nargs="*", default=[128, 128, 16])
    ap.add_argument("--epochs", type=int, default=50)
    a = ap.parse_args()
    import torch  # noqa: F401
    import dorylus_amd as da
    from local_ranks import run_local
    rng = np.random.default_rng(7)
    s, d = rng.integers(0, a.V, a.E // 2).astype(np.uint32), rng.integers(0, a.V, a.E // 2).astype(np.uint32)
    src, dst = np.concatenate([s, d]), np.concatenate([d, s])
    dims, L = a.dims, len(a.dims) - 1
    X = rng.uniform(-1, 1, (a.V, dims[0])).astype(np.float32)
    labels = rng.integers(0, dims[-1], a.V).astype(np.uint32)
    Ws = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(L)]

    def setup(ctx, r, g):
        if g["localVtxCnt"]:
            ctx.upload(0, "x", X[g["localToGlobal"]])
        if g["srcGhostCnt"]:
            ctx.upload(0, "fg", X[g["srcGhost"]].reshape(int(g["srcGhostCnt"]), dims[0]))
        ctx.labels_upload(labels[g["localToGlobal"]])
        for l, W in enumerate(Ws):
            ctx.weight_set(l, "w", W)

    out = {"what": __doc__.split("\n\n")[0], "graph": {"V": a.V, "E": int(len(src)), "dims": dims}, "epochs": a.epochs, "runs": []}
    configs = [("balanced", P, (np.arange(a.V, dtype=np.int64) * P // a.V).astype(np.int32), {"spmm_sweep_cus": 32 // P - 2}) for P in (2, 4)]
    lop = np.zeros(a.V, np.int32)
    lop[int(a.V * 0.98):] = 1
    configs.append(("rank0_holds_98pct", 2, lop, [{"spmm_sweep_cus": 24}, {"spmm_sweep_cus": 4}]))
    # the row gather (K1) beside an exchange: a graph with community structure in community-aligned blocks (what the ldg
    # partitioner recovers, profiles/r06_partition_quality_*.json), 85 % of every row's edges local -- K1's local-first edge
    # split sums them while the ghost rows travel
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_partitions as mp
    csrc, cdst = mp.community_edges(a.V, a.E, 8, 0.85)
    graphs = {"balanced": (src, dst), "rank0_holds_98pct": (src, dst), "k1_community_blocks": (csrc, cdst)}
    configs.append(("k1_community_blocks", 4, (np.arange(a.V, dtype=np.int64) * 4 // a.V).astype(np.int32), {"spmm_variant": 0}))
    # parity of the same path at this size: one epoch of every balanced configuration against the oracle's epoch over the same
    # partitions (every named tensor, max-norm ratio; criteria of tests/helpers.py)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from helpers import oracle_gcn_epoch, rel_err
    out["oracle_check"] = []
    for name, P, parts, share in configs[:2]:
        pobjs = [da.Partition.build(src, dst, parts, r, P) for r in range(P)]
        dl = [(l, nm) for l in range(L) for nm in ("ah",)] + [(l, nm) for l in range(L - 1) for nm in ("h", "aTg")] + [(l, "grad") for l in range(1, L)]
        res = run_local(da, pobjs, parts, dims, da.GCN, 1, setup, dict(share, halo_overlap=1), downloads=dl)
        T, dW = oracle_gcn_epoch(res["views"], parts, X, labels, Ws, a.V)
        worst = 0.0
        for r in range(P):
            for (l, nm), t in res["tensors"][r].items():
                worst = max(worst, rel_err(t, T[r][f"{nm}{l}"]))
            for l in range(L):
                worst = max(worst, rel_err(res["wgrads"][r][l]["w"], dW[l]))
        out["oracle_check"].append({"config": name, "P": P, "max_rel_err_all_named_tensors_and_dW": float(worst), "within_1e-4": bool(worst < 1e-4)})
        sys.stderr.write(json.dumps(out["oracle_check"][-1]) + "\n")
    for name, P, parts, share in configs:
        src, dst = graphs[name]
        bits = {}
        for overlap in (1, 0):
            pobjs = [da.Partition.build(src, dst, parts, r, P) for r in range(P)]
            opts = [dict(sh, halo_overlap=overlap) for sh in share] if isinstance(share, list) else dict(share, halo_overlap=overlap)
            t0 = time.time()
            res = run_local(da, pobjs, parts, dims, da.GCN, a.epochs, setup, opts, timing=True, warm_epochs=2,
                            downloads=[(0, "ah"), (1, "ah"), (0, "aTg")])
            wall = time.time() - t0
            ms = np.stack(res["epoch_ms"])          # [rank][epoch]
            tm = res["timing"]
            rec = {"config": name, "P": P, "halo_overlap": overlap, "options": opts,
                   "vertices_per_rank": [int(v["localVtxCnt"]) for v in res["views"]],
                   "ghost_rows_per_rank": [int(v["srcGhostCnt"]) for v in res["views"]],
                   "epoch_ms_median_max_rank": round(float(np.median(ms.max(axis=0))), 4),
                   "epoch_ms_first_last": [round(float(ms.max(axis=0)[0]), 4), round(float(ms.max(axis=0)[-1]), 4)],
                   "timing_sum_over_ranks_ms": {k: v for k, v in tm.items()},
                   "halo_overlap_fraction": round(tm["halo_hidden"]["ms"] / tm["halo_deferred"]["ms"], 4) if tm["halo_deferred"]["ms"] else None,
                   "spmm_gates": res["gates"], "wall_s": round(wall, 2)}
            out["runs"].append(rec)
            bits[overlap] = res
            sys.stderr.write(json.dumps({k: rec[k] for k in ("config", "P", "halo_overlap", "epoch_ms_median_max_rank", "halo_overlap_fraction", "spmm_gates")}) + "\n")
        same = all(np.array_equal(bits[1]["weights"][r][l]["w"], bits[0]["weights"][r][l]["w"]) for r in range(P) for l in range(L)) and \
            all(np.array_equal(bits[1]["tensors"][r][k], bits[0]["tensors"][r][k]) for r in range(P) for k in bits[1]["tensors"][r])
        out["runs"][-1]["identical_bits_overlap_on_off"] = out["runs"][-2]["identical_bits_overlap_on_off"] = bool(same)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
