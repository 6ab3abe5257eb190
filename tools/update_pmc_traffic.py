#!/usr/bin/env python3
"""Refresh profiles/pmc_traffic.json's K1s entry from a tools/collect_profiles.sh run and stamp it with the git blob hash of
the kernel source it was collected for (bench.py reports `traffic: null` when the stamp and the built source disagree).
  python tools/update_pmc_traffic.py r03b        # reads profiles/<tag>_pmc_fetch_size.txt / _write_size.txt"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import GATMH_STAMP_FILES, source_stamp, spmm_source_stamp  # noqa: E402


def counter_sum(path, counter, pattern):
    tot, names = 0.0, []
    for line in open(path):
        m = re.match(r"(.*?)\s+" + counter + r"\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m and re.search(pattern, m.group(1)):
            tot += float(m.group(3))
            names.append(m.group(1).strip()[:60])
    return tot, names


def main():
    tag = sys.argv[1]
    f = os.path.join(ROOT, "profiles", f"{tag}_pmc_fetch_size.txt")
    w = os.path.join(ROOT, "profiles", f"{tag}_pmc_write_size.txt")
    fetch, names = counter_sum(f, "FETCH_SIZE", r"spmm_sweep_kernel|spmm_sweep_combine")
    write, _ = counter_sum(w, "WRITE_SIZE", r"spmm_sweep_kernel|spmm_sweep_combine")
    launches = 3   # the epoch's aggregations: F=602 forward, F=128 forward, F=128 backward (bench.py --steps 1 --warmup 0)
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    pm = json.load(open(p))
    pm["spmm_variant_2"] = {
        "kernels": "spmm_sweep_kernel<32,10,false,PAIR> (K1s; PAIR on for the five-slab F=602 launch, off for the single-slab F=128 launches)",
        "fetch_size_kb_avg": round(fetch / launches, 1), "write_size_kb_avg": round(write / launches, 1),
        "bytes_per_launch": int((2 * fetch + write) / launches * 1024),
        "source": [f"profiles/{tag}_pmc_fetch_size.txt", f"profiles/{tag}_pmc_write_size.txt"],
        "spmm_hip_blob": spmm_source_stamp(),
        "note": "tools/collect_profiles.sh + tools/update_pmc_traffic.py: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                "`bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-alt`, summed over the sweep kernels of the epoch / 3 launches; "
                "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as reported.  Round 2 (r02e): 10.445 GB",
    }
    # config 4 as one rank of 8 holds it (bench.py key amazon_rank0of8, K1 row gather): FETCH_SIZE of the epoch's five launches
    amz = os.path.join(ROOT, "profiles", f"{tag}_k1_amazon_rank_pmc_fetch_size.txt")
    if not os.path.exists(amz):
        amz = os.path.join(ROOT, "profiles", "r04_k1_amazon_rank_pmc_fetch_size.txt")
    if os.path.exists(amz):
        fetch_a, names_a = counter_sum(amz, "FETCH_SIZE", r"spmm_rows_kernel")
        amz_w = os.path.join(ROOT, "profiles", f"{tag}_k1_amazon_rank_pmc_write_size.txt")
        write_a = counter_sum(amz_w, "WRITE_SIZE", r"spmm_rows_kernel")[0] if os.path.exists(amz_w) else 0.0
        pm["amazon_rank0of8"] = {
            "kernels": "spmm_rows_kernel<32,3> (F=300) + 4 x spmm_rows_kernel<16,1> (F=64): the five aggregations of one epoch",
            "fetch_size_kb_per_epoch": round(fetch_a, 1), "fetch_bytes_per_epoch": int(2 * fetch_a * 1024),
            "write_size_kb_per_epoch": round(write_a, 1), "bytes_per_epoch": int((2 * fetch_a + write_a) * 1024),
            "source": ["profiles/" + os.path.basename(amz)] + (["profiles/" + os.path.basename(amz_w)] if write_a else []),
            "spmm_hip_blob": spmm_source_stamp(),
            "note": "rocprofv3 --pmc FETCH_SIZE pass of `bench.py --workload amazon --emulate 0/8 --steps 1 --warmup 0 --no-cpu-baseline "
                    "--no-alt` (tools/collect_profiles.sh) and, since round 6, the WRITE_SIZE pass of the same command; FETCH_SIZE doubled (gfx950)",
        }
        print("amazon_rank0of8", pm["amazon_rank0of8"]["fetch_bytes_per_epoch"], names_a)
    # config 3 (8-head GAT): the four sweep kernels of one epoch (+ their combine kernels), FETCH and WRITE passes
    gf = os.path.join(ROOT, "profiles", f"{tag}_gatmh_pmc_fetch_size.txt")
    gw = os.path.join(ROOT, "profiles", f"{tag}_gatmh_pmc_write_size.txt")
    if os.path.exists(gf) and os.path.exists(gw):
        pat = r"gatmh_forward_sweep_kernel|gatmh_src_sweep_kernel|gatmh_sweep_combine"
        fetch_g, names_g = counter_sum(gf, "FETCH_SIZE", pat)
        write_g, _ = counter_sum(gw, "WRITE_SIZE", pat)
        pm["gatmh_sweeps"] = {
            "kernels": "gatmh_forward_sweep_kernel + gatmh_src_sweep_kernel, both layers (four edge passes per epoch)",
            "fetch_size_kb_per_epoch": round(fetch_g, 1), "write_size_kb_per_epoch": round(write_g, 1),
            "bytes_per_epoch": int((2 * fetch_g + write_g) * 1024),
            "source": [f"profiles/{tag}_gatmh_pmc_fetch_size.txt", f"profiles/{tag}_gatmh_pmc_write_size.txt"],
            "source_stamp": source_stamp(GATMH_STAMP_FILES),
            "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --gnn gatmh --steps 1 --warmup 0 --no-cpu-baseline --no-alt`; "
                    "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as reported",
        }
        print("gatmh_sweeps", pm["gatmh_sweeps"]["bytes_per_epoch"], names_g)
    json.dump(pm, open(p, "w"), indent=2)
    print(pm["spmm_variant_2"]["bytes_per_launch"], names)


if __name__ == "__main__":
    main()
