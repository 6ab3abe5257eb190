#!/bin/bash
# kernel stats of one bench mode for several option sets: tools/gpu_opts.sh <tag> <gnn> <grep-pattern> "opts1" "opts2" ...   ("-" = no options)
cd /root/repo; export TMPDIR=/tmp
TAG=$1; GNN=$2; PAT=$3; shift 3
mkdir -p gpurun_out/$TAG
i=0
for opts in "$@"; do
  i=$((i+1))
  O=""; [ "$opts" != "-" ] && O="--opt $opts"
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_$i -o k -- python /root/repo/bench.py --gnn $GNN --no-cpu-baseline --no-alt --steps 5 --warmup 1 $O > /tmp/prof_${TAG}_$i.log 2>&1
  python /root/repo/tools/rocprof_summary.py "$(find /tmp/prof_${TAG}_$i -name '*.db' | head -1)" > /root/repo/gpurun_out/$TAG/stats_$i.txt 2>&1
  echo "== $opts"; grep -E "$PAT" /root/repo/gpurun_out/$TAG/stats_$i.txt | cut -c1-60,97-150
  grep -o '"ms_per_step": [0-9.]*' /tmp/prof_${TAG}_$i.log | head -1
done
