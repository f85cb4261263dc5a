#!/bin/bash
# Runs on the GPU box (via gpurun / scripts/gpu_job.sh profile).  rocprofv3 kernel-trace + separate PMC passes for
#   lbvh   every LBVH kernel at 262 k / 2.8 M / 10 M triangles,
#   trace  the trace kernels (per-ray, persistent, dynamic fetch) on the cache-resident 262 k SAH BVH,
#   hbm    the HBM-resident point on the 10 M-triangle LBVH (0.75 GB > 256 MB Infinity Cache): 1080p primary + one AO batch +
#          2^21 incoherent rays with the per-ray kernel, and the incoherent batch alone with kepler_dynamic_fetch (a persistent
#          kernel's grid is the same for every batch, so its per-batch counter means need a run of their own),
#   hair   the same two for hairball-2.8M.
# PMC passes never share a run with another trace domain than --kernel-trace.
# Writes gpurun_out/prof_<tag>/{*.kernels.txt, *.pmc.txt, pmc_summary.json}; copy what is to be judged into profiles/<tag>_*.
# Usage: scripts/profile_round.sh <tag> [lbvh] [trace] [hbm] [hair] [diffuse] [aoframe]
set -u
TAG=${1:-r03}; shift || true
WHAT="${*:-lbvh trace hbm}"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
SUM="python3 scripts/summarize_rocprof.py"

prof_trace() {  # name, workload args...
  local name=$1; shift
  timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/trace -- python3 scripts/workloads.py "$@" > $OUT/$name.trace.log 2>&1
  { echo "== rocprofv3 --kernel-trace --stats -- python3 scripts/workloads.py $* (WL_ONLY=${WL_ONLY:-})"; tail -n 1 $OUT/$name.trace.log; $SUM trace $OUT/$name/trace; } > $OUT/$name.kernels.txt 2>&1
}
prof_pmc() {  # name, "counters", workload args...
  local name=$1; local ctr=$2; shift; shift
  local tagc=$(echo $ctr | tr ' ' '+' | cut -c1-60)
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/$name/pmc_$tagc -- python3 scripts/workloads.py "$@" > $OUT/$name.pmc_$tagc.log 2>&1
}
# (GRBM_GUI_ACTIVE rides along in every pass but the FETCH_SIZE / WRITE_SIZE ones -- the guide wants those alone --: a share is then a
# pass's counts over the SAME pass's cycles, scripts/summarize_rocprof.py `@cycles`)
TRACE_SETS=("FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
  "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" \
  "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" \
  "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" \
  "GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum")

trace_all() {  # name, scene, kernel
  prof_trace $1 trace $2 $3 6
  for C in "${TRACE_SETS[@]}"; do prof_pmc $1 "$C" trace $2 $3 4; done
  $SUM pmc $OUT/$1/pmc_* > $OUT/$1.pmc.txt 2>&1
}

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
    for K in fermi_speculative_while_while tesla_persistent_while_while; do trace_all trace_atrium_$K atrium $K; done ;;
  hbm)
    trace_all trace_courtyard_fermi_speculative_while_while courtyard fermi_speculative_while_while
    WL_ONLY=incoherent; export WL_ONLY
    trace_all trace_courtyard_incoherent_kepler_dynamic_fetch courtyard kepler_dynamic_fetch
    unset WL_ONLY ;;
  hair)
    trace_all trace_hairball_fermi_speculative_while_while hairball fermi_speculative_while_while
    WL_ONLY=incoherent; export WL_ONLY
    trace_all trace_hairball_incoherent_kepler_dynamic_fetch hairball kepler_dynamic_fetch
    unset WL_ONLY ;;
  diffuse)   # BASELINE config 4's frame: the 16 diffuse batches of the hairball frame under the default selector (routed to the persistent body)
    WL_ONLY=diffuse; export WL_ONLY
    trace_all trace_hairball_diffuse_frame_fermi_speculative_while_while hairball fermi_speculative_while_while
    unset WL_ONLY ;;
  aoframe)   # BASELINE config 5's frame: the 16 AO batches of the courtyard-10M frame under the default selector
    WL_ONLY=ao_frame; export WL_ONLY
    trace_all trace_courtyard_ao_frame_fermi_speculative_while_while courtyard fermi_speculative_while_while
    unset WL_ONLY ;;
  esac
done
# what bench.py reads for roofline.binding: {"kernel symbol|grid|counter": mean per dispatch}
F=$(ls $OUT/*atrium*.pmc.txt $OUT/lbvh_*.pmc.txt 2>/dev/null); [ -n "$F" ] && $SUM json $F > $OUT/pmc_summary.json
F=$(ls $OUT/trace_courtyard*.pmc.txt 2>/dev/null | grep -v ao_frame); [ -n "$F" ] && $SUM json $F > $OUT/courtyard10m_pmc_summary.json
F=$(ls $OUT/trace_hairball*.pmc.txt 2>/dev/null | grep -v diffuse_frame); [ -n "$F" ] && $SUM json $F > $OUT/hairball_pmc_summary.json
# whole-frame runs of one launch shape (the persistent symbol has ONE grid for every batch): summaries of their own
F=$(ls $OUT/trace_hairball_diffuse_frame*.pmc.txt 2>/dev/null); [ -n "$F" ] && $SUM json $F > $OUT/hairball_diffuse_frame_pmc_summary.json
F=$(ls $OUT/trace_courtyard_ao_frame*.pmc.txt 2>/dev/null); [ -n "$F" ] && $SUM json $F > $OUT/courtyard_ao_frame_pmc_summary.json
# raw rocprof directories are large: keep the summaries, drop the CSVs of the PMC passes
du -sh $OUT
find $OUT -name "*_counter_collection.csv" -size +8M -delete
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
ls $OUT
