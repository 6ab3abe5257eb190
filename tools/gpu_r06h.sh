#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_gat_mh.py tests/test_gpu_parity.py -q -k "flat_or_peaked or numpy_gnn" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log | cut -c1-400
timeout 900 python tools/local_transport_run.py > $O/local_transport.json 2> $O/local_transport.err; echo "local transport run rc=$?"; tail -9 $O/local_transport.err | cut -c1-400
