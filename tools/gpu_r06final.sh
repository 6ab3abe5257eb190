#!/bin/bash
# round 6 evidence: the profile collection (bench line, kernel summaries, PMC passes) and the 2 / 4 / 8-GPU projections
cd /root/repo; export TMPDIR=/tmp
tools/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1; echo "collect rc=$?"; tail -3 gpurun_out/r06_collect.log
timeout 2400 python tools/scaling_projection.py --out gpurun_out/r06_scaling_projection_a.json --cases reddit:uniform reddit:community amazon:uniform --P 2 4 8 --steps 3 --warmup 1 2> gpurun_out/r06_scaling_projection_a.err; echo "projection a rc=$?"; tail -9 gpurun_out/r06_scaling_projection_a.err
timeout 1500 python tools/scaling_projection.py --out gpurun_out/r06_scaling_projection_b.json --cases amazon:community:block amazon:community:ldg10 --P 8 --steps 3 --warmup 1 2> gpurun_out/r06_scaling_projection_b.err; echo "projection b rc=$?"; tail -3 gpurun_out/r06_scaling_projection_b.err
