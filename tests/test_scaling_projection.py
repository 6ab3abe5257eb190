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
