#!/bin/bash
# round 6, second GPU call: local transport tests again; the 8-head GAT with one statistics gather per BATCH (source side) and
# the sources' scores from the el table (forward); parity of the candidates
cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r06b
timeout 900 python -m pytest tests/test_gpu_local_transport.py -q -x > gpurun_out/r06b/pytest_local.log 2>&1; echo "pytest local rc=$?"; tail -5 gpurun_out/r06b/pytest_local.log
tools/gpu_ab.sh r06b gatmh 'gatmh_(forward|src)_sweep' base aux3 aux3s4 aux3b4 fel fel3
for lib in aux3 fel; do
  DORY_LIB_PATH=/root/repo/build/ab/lib_$lib.so timeout 900 python -m pytest tests/test_gpu_gat_mh.py -q -x > gpurun_out/r06b/pytest_gatmh_$lib.log 2>&1; echo "gatmh parity $lib rc=$?"; tail -3 gpurun_out/r06b/pytest_gatmh_$lib.log
done
