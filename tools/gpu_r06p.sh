#!/bin/bash
R=/root/repo; OUT=$R/gpurun_out; export TMPDIR=/tmp; cd $R
timeout 1500 python -m pytest tests -q -x -m gpu -k "gat and not gat_mh and not gatmh" 2>&1 | tail -3
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_gat_k -o k -- python $R/bench.py --gnn gat --steps 5 --warmup 1 --no-cpu-baseline --no-alt > /tmp/prof_gat_k.log 2>&1
python $R/tools/rocprof_summary.py "$(find /tmp/prof_gat_k -name '*.db' | head -1)" > $OUT/r06_gat_kernel_stats.txt 2>&1
grep -E "colsum|ms_per_step" $OUT/r06_gat_kernel_stats.txt /tmp/prof_gat_k.log | cut -c1-100,130-190 | head -5
grep -o '"ms_per_step": [0-9.]*' /tmp/prof_gat_k.log | head -1
