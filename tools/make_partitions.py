#!/usr/bin/env python3
"""Partition quality for configs 4 / 5 with the repo's OWN partitioner (round 6, review item 3).  CPU only.

    python tools/make_partitions.py [--inside 0.85] [--communities 50] [--P 8] [--tag amazon_community]

Generates the Amazon-size graph (9 430 088 vertices, 231.6 M edge records) with community structure -- bench.py's `community`
generator -- under a seeded SHUFFLE of the vertex ids (a dataset's ids do not arrive METIS-ordered: the partitioner has to
find the communities), writes it as the reference's graph.bsnap (inputs/graphToBinary.cpp:47-160), runs
`dory-inputs partitioner <bsnap> <V> <P> --method=ldg` (the tool of row f-2; the reference calls METIS there,
inputs/partitioner.cpp:113-128 -- same .parts output format) and, for comparison, contiguous blocks of the shuffled ids
(= a random partition with respect to the communities) and the generator's own community order (what an ideal partitioner
would recover).  For each partitioning and rank: local vertices, in-edges, ghost sources / destinations, rows sent to the
busiest peer.  The .parts vectors land in build/parts/<tag>_<method>.npy (int8; they travel to the GPU box for
tools/scaling_projection.py --parts), the statistics in profiles/r06_partition_quality_<tag>.json."""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def community_edges(V, E, C, inside, seed=42):
    """bench.synth_edges('community') with the two knobs exposed (C communities of consecutive ids, `inside` of the edges within)"""
    rng = np.random.default_rng(seed)
    half = E // 2
    s = rng.integers(0, V, half, dtype=np.uint32)
    size = (V + C - 1) // C
    ins = rng.random(half) < inside
    local = (s // size).astype(np.int64) * size + rng.integers(0, size, half)
    d = np.where(ins, np.minimum(local, V - 1), rng.integers(0, V, half)).astype(np.uint32)
    return np.concatenate([s, d]), np.concatenate([d, s])


def shuffled(V, src, dst, seed=7):
    perm = np.random.default_rng(seed).permutation(V).astype(np.uint32)     # new id of old vertex v = perm[v]
    return perm[src], perm[dst], perm


def quality(src, dst, parts, P):
    """per-rank halo statistics of a partitioning (what dataloader.cpp:268-322 would build): ghosts and per-peer send rows"""
    ps, pd = parts[src], parts[dst]
    ranks = []
    for r in range(P):
        m_in = (pd == r)                       # in-edges of rank r's vertices
        rem = m_in & (ps != r)
        gsrc = np.unique(src[rem])             # ghost sources (forward halo rows this rank RECEIVES)
        recv_from = np.bincount(parts[gsrc], minlength=P)
        ranks.append({"rank": r, "vertices": int((parts == r).sum()), "nnz_in": int(m_in.sum()), "remote_in_edges": int(rem.sum()),
                      "ghosts_src": int(len(gsrc)), "recv_rows_from": recv_from.tolist()})
    # rows rank r SENDS to peer q = rows q receives from r (the graph is symmetric: forward and backward lists coincide)
    for r in range(P):
        ranks[r]["send_rows_to"] = [ranks[q]["recv_rows_from"][r] for q in range(P)]
    cut = int((ps != pd).sum())
    return {"edge_cut_records": cut, "edge_cut_frac": round(cut / len(src), 4),
            "ghosts_src_max": max(x["ghosts_src"] for x in ranks), "ghosts_src_mean": round(float(np.mean([x["ghosts_src"] for x in ranks])), 1),
            "send_rows_max_peer": max(max(x["send_rows_to"]) for x in ranks),
            "vertices_max_over_mean": round(max(x["vertices"] for x in ranks) / (len(parts) / P), 4),
            "nnz_in_max_over_mean": round(max(x["nnz_in"] for x in ranks) / (len(src) / P), 4), "ranks": ranks}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=9430088)
    ap.add_argument("--E", type=int, default=231594310)
    ap.add_argument("--communities", type=int, default=50)
    ap.add_argument("--inside", type=float, default=0.85)
    ap.add_argument("--P", type=int, default=8)
    ap.add_argument("--tag", default="amazon_community")
    ap.add_argument("--tmp", default="/tmp")
    ap.add_argument("--methods", nargs="*", default=["ldg"], help="partitioner methods; ldg:N = ldg with N restreaming passes")
    a = ap.parse_args()
    t0 = time.time()
    src, dst = community_edges(a.V, a.E, a.communities, a.inside)
    src, dst, perm = shuffled(a.V, src, dst)
    print(f"generated {len(src)} records in {time.time() - t0:.1f} s", flush=True)
    res = {"what": __doc__.split("\n\n")[0], "V": a.V, "records": int(len(src)), "communities": a.communities, "inside": a.inside, "P": a.P,
           "partitionings": {}}
    outdir = os.path.join(ROOT, "build", "parts")
    os.makedirs(outdir, exist_ok=True)
    parts_by = {}
    parts_by["block"] = (np.arange(a.V, dtype=np.int64) * a.P // a.V).astype(np.int8)          # contiguous blocks of the shuffled ids
    # the generator's own order: old id v sat in community v // size; its new id is perm[v] -> ideal[perm[v]] = block of v
    ideal = np.empty(a.V, np.int8)
    ideal[perm] = (np.arange(a.V, dtype=np.int64) * a.P // a.V).astype(np.int8)
    parts_by["ideal"] = ideal
    work = os.path.join(a.tmp, f"dory_parts_{a.tag}")
    os.makedirs(work, exist_ok=True)
    bs = os.path.join(work, "graph.bsnap")
    t0 = time.time()
    with open(bs, "wb") as f:       # BSHeaderType {int sizeOfVertexType; unsigned numVertices; unsigned long long numEdges} + (src, dst) u32 pairs
        f.write(np.array([4], np.int32).tobytes() + np.array([a.V], np.uint32).tobytes() + np.array([len(src)], np.uint64).tobytes())
        rec = np.empty((len(src), 2), np.uint32)
        rec[:, 0] = src
        rec[:, 1] = dst
        rec.tofile(f)
        del rec
    print(f"wrote {bs} in {time.time() - t0:.1f} s", flush=True)
    for method in a.methods:
        t0 = time.time()
        mname, _, npass = method.partition(":")
        r = subprocess.run([os.path.join(ROOT, "dorylus_amd", "dory-inputs"), "partitioner", bs, str(a.V), str(a.P), f"--method={mname}"] +
                           ([f"--passes={npass}"] if npass else []), cwd=work, capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit(r.stdout + r.stderr)
        secs = time.time() - t0
        p = np.loadtxt(os.path.join(work, f"parts_{a.P}", "graph.bsnap.parts"), dtype=np.int64).astype(np.int8)
        method = method.replace(":", "")
        parts_by[method] = p
        res.setdefault("partitioner_seconds", {})[method] = round(secs, 1)
        print(f"{method}: {secs:.1f} s; {r.stdout.strip().splitlines()[-1]}", flush=True)
    for name, p in parts_by.items():
        np.save(os.path.join(outdir, f"{a.tag}_{name}.npy"), p)
        t0 = time.time()
        res["partitionings"][name] = quality(src, dst, p, a.P)
        q = res["partitionings"][name]
        print(f"{name}: cut {q['edge_cut_frac']}, ghosts max {q['ghosts_src_max']}, busiest peer {q['send_rows_max_peer']} rows, "
              f"balance V {q['vertices_max_over_mean']} nnz {q['nnz_in_max_over_mean']}  ({time.time() - t0:.0f} s)", flush=True)
    os.remove(bs)
    out = os.path.join(ROOT, "profiles", f"r06_partition_quality_{a.tag}.json")
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
