#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_parity.py::test_tanh_matches_libm -q -m gpu 2>&1 | grep -E "^E  |passed|failed" | head -8
python tools/bench_gemm.py 2>&1 | grep -E "NN|TN"
python tools/bench_gemm.py 2>&1 | grep -E "NN|TN"
