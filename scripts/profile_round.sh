#!/bin/bash
# Runs on the GPU box (via gpurun).  rocprofv3 kernel-trace + separate PMC passes for
#   (1) every LBVH kernel at 262 k / 2.8 M / 10 M triangles,
#   (2) the trace kernels (per-ray and persistent) on the cache-resident 262 k SAH BVH,
#   (3) the HBM-resident point: 1080p primary + one AO batch on the 10 M-triangle LBVH (1.3 GB > 256 MB MALL).
# PMC passes never share a run with another trace domain than --kernel-trace.
# Usage: scripts/profile_round.sh <tag> [lbvh] [trace] [hbm]
set -u
TAG=${1:-r02}; shift || true
WHAT="${*:-lbvh trace hbm}"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
SUM="python3 scripts/summarize_rocprof.py"

prof_trace() {  # name, workload args...
  local name=$1; shift
  timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/trace -- python3 scripts/workloads.py "$@" > $OUT/$name.trace.log 2>&1
  { echo "== rocprofv3 --kernel-trace --stats -- python3 scripts/workloads.py $*"; tail -n 1 $OUT/$name.trace.log; $SUM trace $OUT/$name/trace; } > $OUT/$name.kernels.txt 2>&1
}
prof_pmc() {  # name, "counters", workload args...
  local name=$1; local ctr=$2; shift; shift
  local tagc=$(echo $ctr | tr ' ' '+' | cut -c1-60)
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$name/pmc_$tagc -- python3 scripts/workloads.py "$@" > $OUT/$name.pmc_$tagc.log 2>&1
}
TRACE_SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" \
  "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" "TA_BUFFER_WAVEFRONTS_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum")

for W in $WHAT; do
  case $W in
  lbvh)
    for S in atrium hairball courtyard; do
      prof_trace lbvh_$S lbvh $S 8
      for C in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
        prof_pmc lbvh_$S "$C" lbvh $S 4
      done
      $SUM pmc $OUT/lbvh_$S/pmc_* > $OUT/lbvh_$S.pmc.txt 2>&1
    done ;;
  trace)
    for K in fermi_speculative_while_while tesla_persistent_while_while kepler_dynamic_fetch; do
      prof_trace trace_atrium_$K trace atrium $K 6
      for C in "${TRACE_SETS[@]}"; do prof_pmc trace_atrium_$K "$C" trace atrium $K 4; done
      $SUM pmc $OUT/trace_atrium_$K/pmc_* > $OUT/trace_atrium_$K.pmc.txt 2>&1
    done ;;
  hbm)
    K=${PROF_KERNEL:-fermi_speculative_while_while}
    prof_trace trace_courtyard_$K trace courtyard $K 6
    for C in "${TRACE_SETS[@]}"; do prof_pmc trace_courtyard_$K "$C" trace courtyard $K 4; done
    $SUM pmc $OUT/trace_courtyard_$K/pmc_* > $OUT/trace_courtyard_$K.pmc.txt 2>&1 ;;
  esac
done
# raw rocprof directories are large: keep the summaries, drop the CSVs of the PMC passes
du -sh $OUT
find $OUT -name "*_counter_collection.csv" -size +8M -delete
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
ls $OUT
