#!/bin/bash
# final state of round 6: the whole GPU suite, the default bench line, the GAT / 8-head GAT kernel summaries and lines
R=/root/repo; OUT=$R/gpurun_out; export TMPDIR=/tmp; cd $R
timeout 2400 python -m pytest tests/ -q -m gpu -x > $OUT/r06_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/r06_pytest_gpu.log | tail -1
cd /tmp
for g in gat gatmh; do
  rocprofv3 --kernel-trace --stats -d /tmp/prof_${g}_k -o k -- python $R/bench.py --gnn $g --steps 5 --warmup 1 --no-cpu-baseline --no-alt > /tmp/prof_${g}_k.log 2>&1
  python $R/tools/rocprof_summary.py "$(find /tmp/prof_${g}_k -name '*.db' | head -1)" > $OUT/r06_${g}_kernel_stats.txt 2>&1
done
cd $R
python bench.py --gnn gat --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r06_bench_gat.json 2>/dev/null
python bench.py --gnn gatmh --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r06_bench_gatmh.json 2>/dev/null
python bench.py > $OUT/r06_bench.json 2> $OUT/r06_bench.err
python - <<PY
import json
d=json.load(open('$OUT/r06_bench.json')); print(d['ms_per_step'], d['gat']['ms_per_step'], d['gatmh']['ms_per_step'], d['roofline']['traffic'], d['gatmh']['roofline']['traffic'])
PY
