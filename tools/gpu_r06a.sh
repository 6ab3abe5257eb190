#!/bin/bash
# round 6, first GPU call: probes (second-gather forms, stream memory ops), the in-process device transport tests, and the
# 8-head GAT source side with the three forms of its second gather
cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/r06a
(cd tools/probes && timeout 120 ./aux_gather_probe) > gpurun_out/r06a/aux_gather_probe.txt 2>&1; echo "aux probe rc=$?"; cat gpurun_out/r06a/aux_gather_probe.txt
(cd tools/probes && timeout 60 ./waitvalue_probe) > gpurun_out/r06a/waitvalue_probe.txt 2>&1; echo "waitvalue rc=$?"; cat gpurun_out/r06a/waitvalue_probe.txt
timeout 900 python -m pytest tests/test_gpu_local_transport.py -q -x > gpurun_out/r06a/pytest_local.log 2>&1; echo "pytest local rc=$?"; tail -15 gpurun_out/r06a/pytest_local.log
tools/gpu_ab.sh r06a gatmh 'gatmh_(forward|src)_sweep' base aux1 aux2
