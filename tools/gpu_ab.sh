#!/bin/bash
# kernel stats of one bench mode for several A/B libraries: tools/gpu_ab.sh <tag> <gnn> <grep-pattern> lib1 lib2 ...   (libs under build/ab/)
cd /root/repo; export TMPDIR=/tmp
TAG=$1; GNN=$2; PAT=$3; shift 3
mkdir -p gpurun_out/$TAG
for lib in "$@"; do
  export DORY_LIB_PATH=/root/repo/build/ab/lib_$lib.so
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_$lib -o k -- python /root/repo/bench.py --gnn $GNN --no-cpu-baseline --no-alt --steps 5 --warmup 1 > /tmp/prof_${TAG}_$lib.log 2>&1
  python /root/repo/tools/rocprof_summary.py "$(find /tmp/prof_${TAG}_$lib -name '*.db' | head -1)" > /root/repo/gpurun_out/$TAG/stats_$lib.txt 2>&1
  echo "== $lib"; grep -E "$PAT" /root/repo/gpurun_out/$TAG/stats_$lib.txt | cut -c1-60,97-150
  grep -o '"ms_per_step": [0-9.]*' /tmp/prof_${TAG}_$lib.log | head -1
done
