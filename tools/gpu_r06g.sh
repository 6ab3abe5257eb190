#!/bin/bash
# round 6: K1 edge split, per-kernel effect on the Amazon rank; new tests (GAT score gradients, 20k fixture); projection again
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06g; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_gat_mh.py tests/test_gpu_parity.py -q -x -k "flat_or_peaked or numpy_gnn or aggregate_gcn" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-300
cd /tmp
for es in 1 0; do
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_es$es -o k -- python /root/repo/bench.py --workload amazon --emulate 0/8 --steps 5 --warmup 1 --no-cpu-baseline --no-alt --opt spmm_edge_split=$es spmm_blk_force_split=0 > /tmp/prof_es$es.log 2>&1
  python /root/repo/tools/rocprof_summary.py "$(find /tmp/prof_es$es -name '*.db' | head -1)" > /root/repo/$O/amazon_rank_es${es}_kernel_stats.txt 2>&1
  echo "== edge_split=$es"; grep spmm_rows /root/repo/$O/amazon_rank_es${es}_kernel_stats.txt | cut -c1-60,97-150; grep -o '"ms_per_step": [0-9.]*' /tmp/prof_es$es.log | head -1
done
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_es2 -o k -- python /root/repo/bench.py --workload amazon --emulate 0/8 --steps 5 --warmup 1 --no-cpu-baseline --no-alt --opt spmm_edge_split=1 spmm_blk_force_split=1 > /tmp/prof_es2.log 2>&1
python /root/repo/tools/rocprof_summary.py "$(find /tmp/prof_es2 -name '*.db' | head -1)" > /root/repo/$O/amazon_rank_es1_two_launch_kernel_stats.txt 2>&1
echo "== edge_split=1, two launches"; grep spmm_rows /root/repo/$O/amazon_rank_es1_two_launch_kernel_stats.txt | cut -c1-60,97-150; grep -o '"ms_per_step": [0-9.]*' /tmp/prof_es2.log | head -1
