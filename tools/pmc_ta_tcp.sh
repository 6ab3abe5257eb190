#!/bin/bash
# TA / TCP / UTCL1 counter passes (each on its own, kernel trace only, two counters of a block at a time, bounded by
# `timeout`) for a command; appends the rows of kernels matching $1 to $2
#   tools/pmc_ta_tcp.sh <kernel-regex> <out-file> -- <command...>
PAT=$1; OUT=$2; shift 3
export TMPDIR=/tmp
R=$PWD
i=0
: > "$OUT"
for set in "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TCP_TOTAL_ACCESSES_sum TCP_TCP_LATENCY_sum" \
           "TD_TC_STALL_sum TD_TD_BUSY_sum"; do
    i=$((i+1))
    (cd /tmp && timeout 100 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmcx_$$_$i -o p -- "$@" > /tmp/pmcx_$$_$i.log 2>&1) || { echo "pass $i ($set): failed or timed out" >> "$OUT"; continue; }
    python "$R/tools/rocprof_summary.py" "$(find /tmp/pmcx_$$_$i -name '*.db' | head -1)" 2>&1 | grep -E "$PAT" >> "$OUT"
done
