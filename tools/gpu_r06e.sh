#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06e; mkdir -p $O
for m in 0 1 4 8 16 32 0; do echo "== gemm_persistent=$m"; python tools/bench_gemm.py --iters 50 --opt gemm_persistent=$m 2>&1 | grep "NN"; done | tee $O/gemm_persistent_ab.txt
