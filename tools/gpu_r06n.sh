#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06n; mkdir -p $O
timeout 1500 python -m pytest tests -q -x -m gpu -k "gat and not gat_mh and not gatmh" > $O/pytest_gat.log 2>&1; echo "pytest gat rc=$?"; tail -4 $O/pytest_gat.log | cut -c1-300
for m in 1 0; do
  python bench.py --gnn gat --steps 5 --warmup 1 --no-cpu-baseline --opt gat_lazy_edge_tensors=$m > $O/bench_gat_lazy$m.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('$O/bench_gat_lazy$m.json')); print('gat_lazy_edge_tensors=$m', d['ms_per_step'], d['kernel_ms_per_epoch'])
PY
done
