"""The one JSON line the driver reads: `python bench.py` with its defaults (fewer steps here) must print exactly one line
with the contract's keys, the BASELINE metric's workload, a roofline whose numbers follow from each other, and the extra
keys the reviews asked for (R-MAT, community, GAT, 8-head GAT, config 4 as one rank of 8 with its own roofline)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_default_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-rows", "4000"],
                       capture_output=True, text=True, timeout=850, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "edges/s" and d["dtype"] == "f32"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "Reddit GCN 2-layer" in d["config"]["workload"] and d["config"]["layer0_order"].startswith("aggregate-first")
    assert d["config"]["vertices"] == 232965 and abs(d["config"]["edges"] - 114.6e6) < 1e6
    # value = edges of the epoch's three aggregations / epoch time
    assert abs(d["value"] - 3 * d["config"]["edges"] / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and ro["peak"] == 8000.0 and ro["unit"] == "GB/s"
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-4
    assert abs(ro["achieved"] - ro["algorithmic_bytes_per_launch"] / (ro["avg_launch_ms"] * 1e-3) / 1e9) < 0.02 * ro["achieved"]
    assert 0.01 < ro["frac"] < 1.0 and 3 * ro["avg_launch_ms"] < d["ms_per_step"]      # the kernel's time fits inside the step
    lp = ro["l1_path"]
    assert lp["frac"] < lp["frac_of_guide_l2_34.5_TBps"] < lp["frac_of_measured_gather_ceiling_31_TBps"] < 1.0
    assert ro["traffic"] is None or ro["traffic"] > ro["algorithmic_bytes_per_launch"]
    assert d["roofline_gemm"]["bound"] == "mfma" and 0.1 < d["roofline_gemm"]["frac"] < 1.0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["gpu_vs_oracle_rel_err_ah0"] < 1e-5
    assert d["spmm_gates"]["timeouts"] == 0 and d["spmm_gates"]["ungated_launches"] == 0
    for k in ("rmat", "community", "gat", "gatmh", "transform_first", "cached_ah0"):
        assert d[k]["ms_per_step"] > 0, k
    am = d["amazon_rank0of8"]
    assert am["steps"] >= 5 and am["spmm_variant"] == 2 and am["kernel_ms_per_epoch"]["spmm"] > 0
    assert am["partition"]["local_vertices"] == 1178761 and am["partition"]["src_ghosts"] > 7e6
    ar = am["roofline"]
    assert abs(ar["frac"] - ar["achieved"] / 8000.0) < 1e-4 and 0.05 < ar["frac"] < 1.0
    assert 0.3 < ar["gathered_frac_of_achievable_hbm_6.3_TBps"] < 1.0
    assert abs(ar["achieved"] - ar["algorithmic_bytes_per_epoch"] / (ar["aggregation_ms_per_epoch"] * 1e-3) / 1e9) < 0.02 * ar["achieved"]
