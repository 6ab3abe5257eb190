#!/bin/bash
cd /root/repo
for i in 1 2 3 4 5 6 7 8; do python -m pytest "tests/test_gpu_gat_mh.py::test_gat_mh_partitioned_epoch_vs_oracle" -x -q -m gpu 2>&1 | grep -E "passed|failed|AssertionError: \(" | tr '\n' ' '; echo; done
echo "--- whole file"
for i in 1 2 3 4; do python -m pytest tests/test_gpu_gat_mh.py -x -q -m gpu 2>&1 | grep -E "passed|failed|AssertionError: \(" | tr '\n' ' '; echo; done
