#!/bin/bash
# item 6 of the round-4 review: what does keeping the sources' order (no random spread) buy on graphs with structure?
cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r05_locality
for g in community rmat uniform; do
  for opt in "spmm_sweep_layout=3" "spmm_sweep_layout=2" "spmm_sweep_layout=0" "spmm_variant=1" "spmm_variant=0 spmm_order=0"; do
    python bench.py --graph $g --no-cpu-baseline --no-alt --steps 5 --warmup 2 --opt $opt 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$g','$opt',round(d['ms_per_step'],3),d['kernel_ms_per_epoch'],d['spmm_gates']['timeouts'])"
  done
done 2>&1 | tee gpurun_out/r05_locality/epochs.txt
cd /tmp
for opt in "spmm_sweep_layout=3" "spmm_sweep_layout=2"; do
  tag=$(echo $opt | tr '= ' '__')
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_loc_$tag -o p -- python /root/repo/bench.py --graph community --steps 1 --warmup 0 --no-cpu-baseline --no-alt --opt $opt > /tmp/prof_loc_$tag.log 2>&1
  python /root/repo/tools/rocprof_summary.py "$(find /tmp/prof_loc_$tag -name '*.db' | head -1)" > /root/repo/gpurun_out/r05_locality/community_${tag}_pmc_fetch_size.txt 2>&1
  grep "spmm_sweep_kernel.*FETCH_SIZE" /root/repo/gpurun_out/r05_locality/community_${tag}_pmc_fetch_size.txt | cut -c1-50,80-200
done
cd /root/repo
python -m pytest tests/test_gpu_parity.py::test_tanh_matches_libm tests/test_gpu_parity.py::test_gcn_numpy_gnn_fixture -q -m gpu 2>&1 | tail -3
python tools/bench_gemm.py 2>&1 | tail -12
