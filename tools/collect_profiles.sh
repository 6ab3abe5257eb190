#!/bin/bash
# Regenerates the measurement evidence kept under profiles/ on a GPU box (run from the repo root):
#   tools/collect_profiles.sh <tag>        e.g.  gpurun --timeout 1500 -- 'tools/collect_profiles.sh r02'
# Writes gpurun_out/<tag>_*: the bench.py line, the rocprofv3 kernel summary of the same command, and the
# FETCH_SIZE / WRITE_SIZE counter passes (each in its own run, kernel trace only, as MI355X_MICROARCH.md asks).
set -u
TAG=${1:-rXX}
R=$PWD
OUT=$R/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_k -o k -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-alt > /tmp/prof_${TAG}_k.log 2>&1
python "$R/tools/rocprof_summary.py" "$(find /tmp/prof_${TAG}_k -name '*.db' | head -1)" > "$OUT/${TAG}_bench_kernel_stats.txt" 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace -d /tmp/prof_${TAG}_$ctr -o p -- python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-alt > /tmp/prof_${TAG}_$ctr.log 2>&1
    python "$R/tools/rocprof_summary.py" "$(find /tmp/prof_${TAG}_$ctr -name '*.db' | head -1)" > "$OUT/${TAG}_pmc_$(echo $ctr | tr A-Z a-z).txt" 2>&1
done
# further counter passes of the same command, each on its own (kernel trace only): MFMA busy cycles for the GEMMs,
# L2 hit/miss and wave-state counters for the SpMM
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d /tmp/prof_${TAG}_set$i -o p -- python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-alt > /tmp/prof_${TAG}_set$i.log 2>&1
    python "$R/tools/rocprof_summary.py" "$(find /tmp/prof_${TAG}_set$i -name '*.db' | head -1)" > "$OUT/${TAG}_pmc_set$i.txt" 2>&1
done
# config 4 as one rank of 8 holds it (K1 row gather, bench.py key amazon_rank0of8): kernel summary + HBM-side bytes
rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_amz_k -o k -- python "$R/bench.py" --workload amazon --emulate 0/8 --steps 5 --warmup 1 --no-cpu-baseline --no-alt > /tmp/prof_${TAG}_amz_k.log 2>&1
python "$R/tools/rocprof_summary.py" "$(find /tmp/prof_${TAG}_amz_k -name '*.db' | head -1)" > "$OUT/${TAG}_k1_amazon_rank_kernel_stats.txt" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_${TAG}_amz_f -o p -- python "$R/bench.py" --workload amazon --emulate 0/8 --steps 1 --warmup 0 --no-cpu-baseline --no-alt > /tmp/prof_${TAG}_amz_f.log 2>&1
python "$R/tools/rocprof_summary.py" "$(find /tmp/prof_${TAG}_amz_f -name '*.db' | head -1)" > "$OUT/${TAG}_k1_amazon_rank_pmc_fetch_size.txt" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_${TAG}_amz_w -o p -- python "$R/bench.py" --workload amazon --emulate 0/8 --steps 1 --warmup 0 --no-cpu-baseline --no-alt > /tmp/prof_${TAG}_amz_w.log 2>&1
python "$R/tools/rocprof_summary.py" "$(find /tmp/prof_${TAG}_amz_w -name '*.db' | head -1)" > "$OUT/${TAG}_k1_amazon_rank_pmc_write_size.txt" 2>&1
# the community-ordered graph (bench.py --graph community): epoch + HBM-side bytes of its K1s launches
python "$R/bench.py" --graph community --no-cpu-baseline --no-alt > "$OUT/${TAG}_bench_community.json" 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_${TAG}_com_f -o p -- python "$R/bench.py" --graph community --steps 1 --warmup 0 --no-cpu-baseline --no-alt > /tmp/prof_${TAG}_com_f.log 2>&1
python "$R/tools/rocprof_summary.py" "$(find /tmp/prof_${TAG}_com_f -name '*.db' | head -1)" > "$OUT/${TAG}_community_pmc_fetch_size.txt" 2>&1
# config 3 (8-head GAT, sweep forms): kernel summary + HBM-side bytes of its edge passes
rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_gmh_k -o k -- python "$R/bench.py" --gnn gatmh --steps 5 --warmup 1 --no-cpu-baseline --no-alt > /tmp/prof_${TAG}_gmh_k.log 2>&1
python "$R/tools/rocprof_summary.py" "$(find /tmp/prof_${TAG}_gmh_k -name '*.db' | head -1)" > "$OUT/${TAG}_gatmh_kernel_stats.txt" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_${TAG}_gmh_f -o p -- python "$R/bench.py" --gnn gatmh --steps 1 --warmup 0 --no-cpu-baseline --no-alt > /tmp/prof_${TAG}_gmh_f.log 2>&1
python "$R/tools/rocprof_summary.py" "$(find /tmp/prof_${TAG}_gmh_f -name '*.db' | head -1)" > "$OUT/${TAG}_gatmh_pmc_fetch_size.txt" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_${TAG}_gmh_w -o p -- python "$R/bench.py" --gnn gatmh --steps 1 --warmup 0 --no-cpu-baseline --no-alt > /tmp/prof_${TAG}_gmh_w.log 2>&1
python "$R/tools/rocprof_summary.py" "$(find /tmp/prof_${TAG}_gmh_w -name '*.db' | head -1)" > "$OUT/${TAG}_gatmh_pmc_write_size.txt" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/prof_${TAG}_gmh_v -o p -- python "$R/bench.py" --gnn gatmh --steps 1 --warmup 0 --no-cpu-baseline --no-alt > /tmp/prof_${TAG}_gmh_v.log 2>&1
python "$R/tools/rocprof_summary.py" "$(find /tmp/prof_${TAG}_gmh_v -name '*.db' | head -1)" > "$OUT/${TAG}_gatmh_pmc_valu.txt" 2>&1
# the reference's single-head GAT prototype (rows a-6 / a-8): kernel summary of its epoch
rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_gat_k -o k -- python "$R/bench.py" --gnn gat --steps 5 --warmup 1 --no-cpu-baseline --no-alt > /tmp/prof_${TAG}_gat_k.log 2>&1
python "$R/tools/rocprof_summary.py" "$(find /tmp/prof_${TAG}_gat_k -name '*.db' | head -1)" > "$OUT/${TAG}_gat_kernel_stats.txt" 2>&1
cd "$R"
python bench.py --gnn gat --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_bench_gat.json" 2>/dev/null
python bench.py --gnn gatmh --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/${TAG}_bench_gatmh.json" 2>/dev/null
python tools/bench_small.py > "$OUT/${TAG}_bench_small.txt" 2>/dev/null
ls -la "$OUT" | grep "${TAG}_"
