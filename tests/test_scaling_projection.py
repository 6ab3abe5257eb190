"""tools/scaling_projection.py: the model that turns per-rank measurements into a projected multi-GPU epoch (CPU only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _rank(r, compute, hide, rows_to, gs, gd, nnz):
    return {"rank": r, "compute_ms": compute, "local_first_ms_per_aggregate": hide, "send_rows_fwd": rows_to, "send_rows_bwd": rows_to,
            "ghosts_src": gs, "ghosts_dst": gd, "nnz_in": nnz, "nnz_out": nnz}


def test_projection_model_hidden_and_exposed_exchanges():
    import scaling_projection as sp
    dims = [602, 128, 41]
    # two ranks, 100 000 rows to the peer at 128 floats: 51.2 MB per exchange -> 0.335 ms on one 153 GB/s link
    fast = [_rank(0, 10.0, 5.0, [0, 100000], 100000, 100000, 1000), _rank(1, 9.0, 5.0, [100000, 0], 100000, 100000, 900)]
    p = sp.project(fast, dims, 2)
    assert p["exposed_halo_ms_max"] == 0.0                      # hidden under a 5 ms local-source launch
    assert abs(p["projected_epoch_ms"] - (10.0 + p["allreduce_ms"])) < 1e-3 and p["slowest_rank"] == 0
    assert p["halo_bytes_per_exchange_max_peer"] == 100000 * 128 * 4
    assert abs(p["nnz_in_max_over_mean"] - 1000 / 950) < 1e-3
    ex = p["per_rank"][0]["exchanges"]
    assert len(ex) == 2 and ex[0]["dir"] == "fwd" and ex[1]["dir"] == "bwd" and abs(ex[0]["exchange_ms"] - (0.3346 + 0.0410 + 0.03)) < 2e-3   # link + pack/unpack of 200 000 rows at the measured 5 TB/s + latency
    # nothing to hide under: every exchange is exposed in full
    slow = [_rank(0, 10.0, 0.0, [0, 100000], 100000, 100000, 1000), _rank(1, 9.0, 0.0, [100000, 0], 100000, 100000, 900)]
    q = sp.project(slow, dims, 2)
    assert abs(q["exposed_halo_ms_max"] - 2 * ex[0]["exchange_ms"]) < 1e-3
    # a 3-layer model exchanges twice per direction
    r3 = sp.project(fast, [300, 64, 64, 25], 2)
    assert len(r3["per_rank"][0]["exchanges"]) == 4


def test_bench_reads_the_committed_projection():
    sys.path.insert(0, ROOT)
    import bench
    pj = bench.scaling_projection_for("reddit", "uniform", 3)    # no 3-way projection is made
    assert pj["available"] is False and "why" in pj
    path = os.path.join(ROOT, "profiles", "r05_scaling_projection.json")
    if os.path.exists(path):
        for P in (2, 4, 8):
            e = bench.scaling_projection_for("reddit", "uniform", P)
            assert e["available"] and e["projected_epoch_ms"] > 0 and e["model"]["link_GBps"] == 153.0


def test_pmc_traffic_stamp_follows_code_not_comments(tmp_path, monkeypatch):
    """bench.py's roofline.traffic is only reported while profiles/pmc_traffic.json was collected for the kernels that are
    built: the stamp hashes spmm.hip + sweep_core.hpp with comments and white space removed.  The committed figure must carry
    the stamp of the committed sources (else the driver's line would say `traffic: null`), a reworded comment must not
    orphan it, a code change must."""
    import json
    import shutil
    import bench
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        d = json.load(f)
    assert d["spmm_variant_2"]["spmm_hip_blob"] == bench.spmm_source_stamp()
    # config 3's sweep kernels (round 6: gatmh.roofline.traffic) are stamped with gat_mh_sweep.hip + sweep_core.hpp
    assert d["gatmh_sweeps"]["source_stamp"] == bench.source_stamp(bench.GATMH_STAMP_FILES)
    assert d["gatmh_sweeps"]["bytes_per_epoch"] > 0 and d["amazon_rank0of8"]["bytes_per_epoch"] > d["amazon_rank0of8"]["fetch_bytes_per_epoch"]
    csrc = tmp_path / "dorylus_amd" / "csrc"
    csrc.mkdir(parents=True)
    for f in ("spmm.hip", "sweep_core.hpp"):
        shutil.copy(os.path.join(ROOT, "dorylus_amd", "csrc", f), csrc / f)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    base = bench.spmm_source_stamp()
    with open(csrc / "spmm.hip", "a") as f:
        f.write("\n// a remark\n/* and a block\n   of them */\n")
    assert bench.spmm_source_stamp() == base
    with open(csrc / "spmm.hip", "a") as f:
        f.write("\nstatic int one_more_symbol;\n")
    assert bench.spmm_source_stamp() != base


def test_partition_quality_counts_ghosts_and_peer_rows():
    """tools/make_partitions.py: per-rank ghosts and per-peer rows of a partitioning, on a graph small enough to count by hand"""
    import numpy as np
    import make_partitions as mp
    # 6 vertices, two ranks {0,1,2} / {3,4,5}; symmetric records
    e = [(0, 1), (1, 2), (2, 3), (3, 4), (0, 5), (1, 5)]
    src = np.array([a for a, b in e] + [b for a, b in e], np.uint32)
    dst = np.array([b for a, b in e] + [a for a, b in e], np.uint32)
    parts = np.array([0, 0, 0, 1, 1, 1], np.int8)
    q = mp.quality(src, dst, parts, 2)
    r0, r1 = q["ranks"]
    assert r0["vertices"] == 3 and r0["ghosts_src"] == 2 and r0["recv_rows_from"] == [0, 2]      # rank 0 gathers 3 and 5
    assert r1["ghosts_src"] == 3 and r1["recv_rows_from"] == [3, 0]                               # rank 1 gathers 0, 1, 2
    assert r0["send_rows_to"] == [0, 3] and r1["send_rows_to"] == [2, 0]
    assert q["edge_cut_records"] == 6 and q["send_rows_max_peer"] == 3 and q["ghosts_src_max"] == 3
    # the community generator keeps `inside` of its edges within a community of consecutive ids, and the shuffle is a relabelling
    s, d = mp.community_edges(1000, 20000, 10, 0.9)
    assert abs(((s // 100) == (d // 100)).mean() - 0.91) < 0.02
    s2, d2, perm = mp.shuffled(1000, s, d)
    assert np.array_equal(perm[s], s2) and sorted(perm.tolist()) == list(range(1000))


def test_reproject_recomputes_from_stored_rank_measurements(tmp_path):
    """--reproject: the committed projection is a pure function of its per-rank records and the model constants"""
    import json
    import shutil
    import scaling_projection as sp
    src = os.path.join(ROOT, "profiles", "r06_scaling_projection.json")
    if not os.path.exists(src):
        return
    dst = tmp_path / "p.json"
    shutil.copy(src, dst)
    before = json.load(open(src))
    after = sp.reproject(str(dst))
    for c, r in before["cases"].items():
        for P, v in r["by_P"].items():
            assert abs(after["cases"][c]["by_P"][P]["projected_epoch_ms"] - v["projected_epoch_ms"]) < 1e-6, (c, P)
