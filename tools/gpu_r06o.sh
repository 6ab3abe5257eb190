#!/bin/bash
R=/root/repo; OUT=$R/gpurun_out; export TMPDIR=/tmp; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "lazy_edge or gat_engine" 2>&1 | tail -3
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_gat_k -o k -- python $R/bench.py --gnn gat --steps 5 --warmup 1 --no-cpu-baseline --no-alt > /tmp/prof_gat_k.log 2>&1
python $R/tools/rocprof_summary.py "$(find /tmp/prof_gat_k -name '*.db' | head -1)" > $OUT/r06_gat_kernel_stats.txt 2>&1
cd $R
python bench.py --gnn gat --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r06_bench_gat.json 2>/dev/null
python bench.py > $OUT/r06_bench.json 2> $OUT/r06_bench.err
python - <<PY
import json
d=json.load(open('$OUT/r06_bench.json')); print(d['ms_per_step'], d['gat']['ms_per_step'], d['gat'].get('edge_sweeps_per_epoch'), d['gatmh']['ms_per_step'], d['roofline']['traffic'])
PY
