#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gat_mh.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-alt --steps 10 --warmup 2 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('gcn', d['ms_per_step'], d['kernel_ms_per_epoch'])"
python tools/bench_spmm.py --F 602 128 --variants 2 --groups 32 --slabs 0 2>&1 | grep "F="
bash tools/gpu_r05c.sh r05f 'gatmh_sweep_rows=6 spmm_sweep_window_kb=3584' 'gatmh_sweep_rows=8 spmm_sweep_window_kb=3584' 2>&1 | grep -E "==|forward_sweep|ms_per_step"
