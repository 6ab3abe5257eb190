#!/bin/bash
# re-collect the 8-head GAT's PMC passes (gat_mh_sweep.hip changed textually: dead forms removed)
R=/root/repo; OUT=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  l=$(echo $ctr | tr A-Z a-z)
  rocprofv3 --pmc $ctr --kernel-trace -d /tmp/prof_gmh_$ctr -o p -- python $R/bench.py --gnn gatmh --steps 1 --warmup 0 --no-cpu-baseline --no-alt > /tmp/prof_gmh_$ctr.log 2>&1
  python $R/tools/rocprof_summary.py "$(find /tmp/prof_gmh_$ctr -name '*.db' | head -1)" > $OUT/r06_gatmh_pmc_$l.txt 2>&1
done
rocprofv3 --kernel-trace --stats -d /tmp/prof_gmh_k -o k -- python $R/bench.py --gnn gatmh --steps 5 --warmup 1 --no-cpu-baseline --no-alt > /tmp/prof_gmh_k.log 2>&1
python $R/tools/rocprof_summary.py "$(find /tmp/prof_gmh_k -name '*.db' | head -1)" > $OUT/r06_gatmh_kernel_stats.txt 2>&1
grep -E "gatmh_(forward|src)_sweep" $OUT/r06_gatmh_kernel_stats.txt | cut -c1-60,97-150
