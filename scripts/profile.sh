#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats, then separate PMC passes
# (FETCH_SIZE and WRITE_SIZE in their own passes, MI355X_MICROARCH "rocprofv3 PMC slots").
# Usage: scripts/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-graph $*"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 bench.py $ARGS > $OUT/bench_pmc_$N.log 2>&1
done
python3 scripts/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
