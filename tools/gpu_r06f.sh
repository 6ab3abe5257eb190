#!/bin/bash
# round 6: K1's edge-split overlap (local-source edges of every row first): parity, local transport, projections
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06f; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_local_transport.py tests/test_gpu_multirank.py -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for es in 1 0; do
  timeout 600 python bench.py --workload amazon --emulate 0/8 --steps 5 --warmup 1 --no-cpu-baseline --no-alt --opt spmm_edge_split=$es > $O/amazon_rank_es$es.json 2> $O/amazon_rank_es$es.err; echo "amazon es=$es rc=$?"
  python - <<PY
import json
d=json.load(open('$O/amazon_rank_es$es.json')); print('edge_split $es', d['ms_per_step'], d['kernel_ms_per_epoch'])
PY
done
timeout 2000 python tools/scaling_projection.py --out $O/scaling_projection.json --cases amazon:community:block amazon:community:ldg10 amazon:uniform --P 8 --steps 3 --warmup 1 2> $O/scaling_projection.err; echo "projection rc=$?"; tail -4 $O/scaling_projection.err
