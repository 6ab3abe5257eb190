#!/bin/bash
# round 6, fourth GPU call: K2 persistent form (test + A/B), 8-head GAT with six rows per group, local transport with per-rank CU shares
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "gemm" > $O/pytest_gemm.log 2>&1; echo "pytest gemm rc=$?"; tail -4 $O/pytest_gemm.log
for m in 0 1 0 1; do echo "== gemm_persistent=$m"; python tools/bench_gemm.py --iters 50 --opt gemm_persistent=$m 2>&1 | grep TFLOP; done | tee $O/gemm_persistent_ab.txt
for m in 0 1; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alt --opt gemm_persistent=$m > $O/bench_gemm_p$m.json 2> $O/bench_gemm_p$m.err
  python - <<PY
import json
d=json.load(open('$O/bench_gemm_p$m.json')); print('bench gemm_persistent=$m', d['ms_per_step'], d['kernel_ms_per_epoch'])
PY
done
cd /tmp
export DORY_LIB_PATH=/root/repo/build/ab/lib_fr6.so
for r in 0 6; do
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_fr6_$r -o k -- python /root/repo/bench.py --gnn gatmh --no-cpu-baseline --no-alt --steps 5 --warmup 1 --opt gatmh_sweep_rows=$r > /tmp/prof_fr6_$r.log 2>&1
  python /root/repo/tools/rocprof_summary.py "$(find /tmp/prof_fr6_$r -name '*.db' | head -1)" > /root/repo/$O/stats_fr6_rows$r.txt 2>&1
  echo "== fr6 gatmh_sweep_rows=$r"; grep -E 'gatmh_(forward|src)_sweep' /root/repo/$O/stats_fr6_rows$r.txt | cut -c1-60,97-150; grep -o '"ms_per_step": [0-9.]*' /tmp/prof_fr6_$r.log | head -1
done
unset DORY_LIB_PATH
cd /root/repo
timeout 900 python tools/local_transport_run.py > $O/local_transport.json 2> $O/local_transport.err; echo "local transport run rc=$?"; tail -6 $O/local_transport.err
