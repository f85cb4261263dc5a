#!/bin/bash
# PMC before / after of the tail hand-off: rocprofv3 --pmc passes (one counter set per run, kernel-trace only beside
# them) of scripts/studies/handoff_workload.py with the hand-off off and on.  usage: handoff_pmc.sh <tag> <scene> <batch> <K> <T> <A>
set -u
TAG=$1; SCENE=$2; BATCH=$3; K=$4; T=$5; A=$6
OUT=gpurun_out/$TAG/pmc_${SCENE}_${BATCH}; mkdir -p $OUT
export TMPDIR=/tmp
SETS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
      "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
FLAGS=0; [ "$K" = "1" ] && FLAGS=2
for MODE in off on; do
  export NTR_TRACE_MINIPOOL=$K NTR_TRACE_HANDOFF_BELOW=$T NTR_TRACE_HANDOFF_KEEP_WAVES=$A NTR_TRACE_HANDOFF_FLAGS=$FLAGS
  if [ $MODE = off ]; then export NTR_TRACE_HANDOFF=0; else export NTR_TRACE_HANDOFF=1; fi
  i=0
  for C in "${SETS[@]}"; do
    timeout -k 5 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$MODE/set$i -- python3 scripts/studies/handoff_workload.py $SCENE $BATCH 6 > $OUT/$MODE.set$i.log 2>&1
    i=$((i+1))
  done
  { echo "== hand-off $MODE: K=$K T=$T A=$A, $SCENE $BATCH"; tail -n 1 $OUT/$MODE.set0.log | cut -c1-600; python3 scripts/summarize_rocprof.py pmc $OUT/$MODE/set*; } > $OUT/$MODE.pmc.txt 2>&1
done
find $OUT -name "*.csv" -size +4M -delete
grep -h "perray_mini\|== hand" $OUT/off.pmc.txt $OUT/on.pmc.txt | cut -c1-220
