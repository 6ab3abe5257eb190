#!/bin/bash
# tools/bench_gemm.py against several A/B libraries: tools/gpu_ab_gemm.sh "<bench_gemm args>" lib1 lib2 ...
cd /root/repo
ARGS=$1; shift
for lib in "$@"; do
  echo "== $lib"
  DORY_LIB_PATH=/root/repo/build/ab/lib_$lib.so python tools/bench_gemm.py $ARGS 2>&1 | grep TFLOP
done
