#!/bin/bash
# kernel stats of the gatmh epoch for several option sets: tools/gpu_r05c.sh tag "opts1" "opts2" ...
cd /root/repo; export TMPDIR=/tmp
TAG=$1; shift
mkdir -p gpurun_out/$TAG
i=0
for opts in "$@"; do
  i=$((i+1))
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_$i -o k -- python /root/repo/bench.py --gnn gatmh --no-cpu-baseline --no-alt --steps 5 --warmup 1 --opt $opts > /tmp/prof_${TAG}_$i.log 2>&1
  python /root/repo/tools/rocprof_summary.py "$(find /tmp/prof_${TAG}_$i -name '*.db' | head -1)" > /root/repo/gpurun_out/$TAG/stats_$i.txt 2>&1
  echo "== $opts"; grep -E "gatmh_(forward|bwd|src)_(sweep|dst|src|blocked)|gatmh_forward_blocked" /root/repo/gpurun_out/$TAG/stats_$i.txt | cut -c1-60,97-150
  grep -o '"ms_per_step": [0-9.]*' /tmp/prof_${TAG}_$i.log | head -1
done
