#!/bin/bash
# GAT-MH sweep: parity + per-kernel stats of the gatmh epoch (args: tag, extra bench opts)
cd /root/repo; export TMPDIR=/tmp
TAG=${1:-r05b}; shift
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_gpu_gat_mh.py -x -q -m gpu > gpurun_out/$TAG/pytest_gatmh.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/$TAG/pytest_gatmh.log
for i in 1 2 3; do python tools/debug/gatmh_part_debug.py 0 0 2>&1 | grep " 1 o "; done
timeout 600 python bench.py --gnn gatmh --no-cpu-baseline --no-alt --steps 10 --warmup 2 "$@" > gpurun_out/$TAG/bench_gatmh.json 2> gpurun_out/$TAG/bench_gatmh.err; echo "bench gatmh rc=$?"
python -c "import json;d=json.load(open('gpurun_out/$TAG/bench_gatmh.json'));print(d['ms_per_step'], d['kernel_ms_per_epoch'], d.get('spmm_gates'))"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o k -- python /root/repo/bench.py --gnn gatmh --no-cpu-baseline --no-alt --steps 5 --warmup 1 "$@" > /tmp/prof_$TAG.log 2>&1
python /root/repo/tools/rocprof_summary.py "$(find /tmp/prof_$TAG -name '*.db' | head -1)" > /root/repo/gpurun_out/$TAG/gatmh_kernel_stats.txt 2>&1
head -24 /root/repo/gpurun_out/$TAG/gatmh_kernel_stats.txt | cut -c1-150
