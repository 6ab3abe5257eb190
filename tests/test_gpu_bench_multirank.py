"""bench.py's N > 1 path end to end on ONE GPU: `--transport host` runs the same script the driver launches for the scaling
curve -- contiguous partitions per rank, halo self-check (received ghost rows bit-equal to the owners' rows, overlapped and
sequential schedules bit-identical), K timed epochs between barriers, MAX over ranks, whole-job edges/s, the multi_gpu
bookkeeping -- with 2 and 3 ranks as processes sharing device 0; only the bytes travel through gloo callbacks instead of
RCCL (two ranks cannot share a GPU under RCCL).  What it cannot cover: the ~40 lines of ncclSend / ncclRecv / ncclAllReduce."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3, 4])
def test_bench_multirank_dry_run_on_one_gpu(world):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--transport", "host",
           "--device", "0", "--steps", "2", "--warmup", "1", "--scale", "0.08", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["halo_selfcheck"] is True                            # ghost rows bit-equal, both schedules bit-identical
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["unit"] == "edges/s"
    m = d["multi_gpu"]
    assert len(m["vertices_per_rank"]) == world and sum(m["vertices_per_rank"]) == d["config"]["vertices"]
    assert m["halo_bytes_received_per_epoch_total"] > 0 and m["nnz_max_over_mean"] >= 1.0
    assert m["spmm_gate_timeouts_per_rank"] is not None and len(m["spmm_ungated_launches_per_rank"]) == world
    assert "host callbacks" in d["config"]["transport"]
    assert d["roofline"] is not None and d["roofline"]["traffic"] is None      # (the PMC traffic figure is the 1-GPU run's)
    # round 4, first-contact hardening: (a) the preflight ran its three collectives and verified them, (b) the warm-up
    # sampled the CUs left to the exchange's kernels, (c) the overlap of the exchanges with the local-source launches is
    # reported per rank
    pf = m["preflight"]
    assert pf["ok"] is True and pf["bytes_per_peer"] == 64 * 4096 * 4 and pf["allreduce_bytes"] == 19 * 4096 * 4
    assert pf["halo_forward_s"] >= 0 and pf["allreduce_adam_s"] >= 0 and "gloo" in pf["transport"]
    rp = m["reserve_probe"]
    assert rp["chosen"] in (2, 4, 8) and rp["tried"][0]["reserve_cus"] == 2
    assert m["spmm_sweep_reserve_cus"] == rp["chosen"]
    fr = m["halo_overlap_fraction_per_rank"]
    assert len(fr) == world and all(f is None or 0.0 <= f <= 1.0 for f in fr)
    assert m["halo_deferred_ms_per_epoch_max_rank"] >= 0 and m["spmm_beside_halo_ms_per_epoch_max_rank"] >= 0
    # round 5: the written expectation travels with the record (projection of tools/scaling_projection.py for this workload,
    # graph and world size when one is on file: world 2 and 4 are; 3 is not and must say so)
    pj = m["projected"]
    assert isinstance(pj, dict) and "available" in pj
    if pj["available"]:
        assert pj["projected_epoch_ms"] > 0 and pj["compute_ms_max_rank"] > 0 and pj["model"]["link_GBps"] == 153.0
    else:
        assert "why" in pj


def test_bench_multirank_gat_dry_run_checks_its_halo():
    """--gnn gat with N > 1: the halo self-check (z forward into fg_z, grad backward into bg_d, bit-equal to the owners' rows)
    runs before anything is timed (round 5; it was skipped for the GAT orders before)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "host", "--gnn", "gat",
           "--device", "0", "--steps", "2", "--warmup", "1", "--scale", "0.08", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["halo_selfcheck"] is True and d["ms_per_step"] > 0
