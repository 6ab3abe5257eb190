#!/bin/bash
# round 6, third GPU call: 8-head GAT with six rows per 32-lane group; the local transport evidence run; Amazon rank K1 with the
# rows ordered by median source id + the WRITE_SIZE pass; the P = 8 projection of the community graph under two partitionings
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r06c; mkdir -p $O
tools/gpu_ab.sh r06c gatmh 'gatmh_(forward|src)_sweep' cur fr6 fr6sr6
DORY_LIB_PATH=/root/repo/build/ab/lib_fr6.so timeout 900 python -m pytest tests/test_gpu_gat_mh.py -q -x > $O/pytest_gatmh_fr6.log 2>&1; echo "gatmh parity fr6 rc=$?"; tail -2 $O/pytest_gatmh_fr6.log
timeout 900 python tools/local_transport_run.py > $O/local_transport.json 2> $O/local_transport.err; echo "local transport run rc=$?"; cat $O/local_transport.err | tail -8
for ord in 1 3; do
  timeout 600 python bench.py --workload amazon --emulate 0/8 --steps 5 --warmup 1 --no-cpu-baseline --no-alt --opt spmm_order=$ord > $O/amazon_rank_order$ord.json 2> $O/amazon_rank_order$ord.err; echo "amazon order $ord rc=$?"
  python - <<PY
import json
d=json.load(open('$O/amazon_rank_order$ord.json')); print('order $ord', d['ms_per_step'], d['kernel_ms_per_epoch'])
PY
done
cd /tmp
for ord in 1 3; do
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_amz_f$ord -o p -- python /root/repo/bench.py --workload amazon --emulate 0/8 --steps 1 --warmup 0 --no-cpu-baseline --no-alt --opt spmm_order=$ord > /tmp/prof_amz_f$ord.log 2>&1
  python /root/repo/tools/rocprof_summary.py "$(find /tmp/prof_amz_f$ord -name '*.db' | head -1)" > /root/repo/$O/k1_amazon_rank_order${ord}_pmc_fetch_size.txt 2>&1
done
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_amz_w -o p -- python /root/repo/bench.py --workload amazon --emulate 0/8 --steps 1 --warmup 0 --no-cpu-baseline --no-alt > /tmp/prof_amz_w.log 2>&1
python /root/repo/tools/rocprof_summary.py "$(find /tmp/prof_amz_w -name '*.db' | head -1)" > /root/repo/$O/k1_amazon_rank_pmc_write_size.txt 2>&1
grep -h spmm_rows /root/repo/$O/k1_amazon_rank_*pmc*.txt | cut -c1-50,90-160
cd /root/repo
timeout 1500 python tools/scaling_projection.py --out $O/scaling_projection_community.json --cases amazon:community:block amazon:community:ldg10 --P 8 --steps 3 --warmup 1 2> $O/scaling_projection.err; echo "projection rc=$?"; tail -4 $O/scaling_projection.err
