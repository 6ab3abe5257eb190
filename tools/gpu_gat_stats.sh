cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_gat -o k -- python /root/repo/bench.py --gnn gat --no-cpu-baseline --no-alt --steps 5 --warmup 1 > /tmp/prof_gat.log 2>&1
python /root/repo/tools/rocprof_summary.py "$(find /tmp/prof_gat -name '*.db' | head -1)" > /root/repo/gpurun_out/r05_gat_kernel_stats.txt 2>&1
head -30 /root/repo/gpurun_out/r05_gat_kernel_stats.txt | cut -c1-70,97-150
