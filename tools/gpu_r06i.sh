#!/bin/bash
cd /tmp; export TMPDIR=/tmp
O=/root/repo/gpurun_out/r06i; mkdir -p $O
for kb in 0 2560 3072 3584 4096; do
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_w$kb -o k -- python /root/repo/bench.py --gnn gatmh --no-cpu-baseline --no-alt --steps 5 --warmup 1 --opt gatmh_src_window_kb=$kb > /tmp/prof_w$kb.log 2>&1
  python /root/repo/tools/rocprof_summary.py "$(find /tmp/prof_w$kb -name '*.db' | head -1)" > $O/stats_srcwin$kb.txt 2>&1
  echo "== gatmh_src_window_kb=$kb"; grep -E 'gatmh_(forward|src)_sweep' $O/stats_srcwin$kb.txt | cut -c1-60,97-150; grep -o '"ms_per_step": [0-9.]*' /tmp/prof_w$kb.log | head -1
done
