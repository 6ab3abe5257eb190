#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
for lib in bm128 bm256 bm128 bm256; do
  echo "== $lib: Amazon layer 0 (300 -> 64, N = 1178761)"; DORY_LIB_PATH=/root/repo/build/ab/lib_$lib.so python tools/bench_gemm.py --N 1178761 --dims 300 64 25 --iters 30 2>&1 | grep TFLOP
  echo "== $lib: Reddit layer 1 (128 -> 41, N = 232965)"; DORY_LIB_PATH=/root/repo/build/ab/lib_$lib.so python tools/bench_gemm.py --N 232965 --dims 128 41 7 --iters 50 2>&1 | grep TFLOP
done
