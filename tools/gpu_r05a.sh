#!/bin/bash
# round 5, first contact of the GAT-MH sweep forward: parity, per-kernel times, headline regression check of the refactor
cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests/test_gpu_gat_mh.py -x -q -m gpu > gpurun_out/r05a/pytest_gatmh.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05a/pytest_gatmh.log
tail -5 gpurun_out/r05a/pytest_gatmh.log
timeout 600 python bench.py --gnn gatmh --no-cpu-baseline --no-alt --steps 10 --warmup 2 > gpurun_out/r05a/bench_gatmh.json 2> gpurun_out/r05a/bench_gatmh.err; echo "bench gatmh rc=$?"
cut -c1-600 gpurun_out/r05a/bench_gatmh.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_gatmh -o gatmh -- python /root/repo/bench.py --gnn gatmh --no-cpu-baseline --no-alt --steps 10 --warmup 2 > /dev/null 2>&1)
f=$(find /tmp/prof_gatmh -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05a/gatmh_kernel_stats.csv; head -25 "$f" | cut -c1-220
timeout 600 python bench.py --no-cpu-baseline --no-alt --steps 10 --warmup 2 > gpurun_out/r05a/bench_gcn.json 2> gpurun_out/r05a/bench_gcn.err; echo "bench gcn rc=$?"
cut -c1-400 gpurun_out/r05a/bench_gcn.json
timeout 300 python tools/bench_spmm.py --F 602 128 --variants 2 --groups 32 --slabs 0 > gpurun_out/r05a/bench_spmm.log 2>&1; tail -8 gpurun_out/r05a/bench_spmm.log
