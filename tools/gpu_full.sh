#!/bin/bash
# the full GPU suite + the default bench line (what the driver runs at round end): tools/gpu_full.sh <tag>
cd /root/repo; export TMPDIR=/tmp
TAG=${1:-full}
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests/ -q -m gpu -x > gpurun_out/$TAG/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/$TAG/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/$TAG/bench.json'))
print('headline', d['ms_per_step'], d['value'], d['roofline'])
for k in ('rmat','gat','gatmh','transform_first','cached_ah0','amazon_rank0of8'):
    v=d.get(k)
    if isinstance(v,dict): print(k, v.get('ms_per_step'), v.get('roofline'))
    else: print(k, v)
print('cpu', d.get('cpu_baseline'))
PY
