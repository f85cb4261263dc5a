#!/bin/bash
# pair_pmc.sh <tag>: PMC counters of the bench step's AO launches with one ray per lane (trace_bvh_perray<1, false, true, true>) and with two
# (The kernel lives in scripts/studies/rejected_patches/two_rays_per_lane.patch: apply it, `make -C ntrace_amd/csrc`, then run this.)
# (trace_bvh_perray_pair, NTR_TRACE_PAIR=1) -- wait share, VALU / TA load, waves -- for the record of the two-rays-per-lane experiment
# (VERDICT r04 item 1).  Separate --pmc passes, kernel trace only beside them.
TAG=${1:-r05}
OUT=gpurun_out/pair_pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-cold-order"
for MODE in 0 1; do
  export NTR_TRACE_PAIR=$MODE
  i=0
  for C in "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM"; do
    timeout -k 5 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pair$MODE/set$i -- python3 bench.py $ARGS > $OUT/pair$MODE.set$i.log 2>&1
    i=$((i+1))
  done
  python3 scripts/summarize_rocprof.py pmc $OUT/pair$MODE/set* | grep "trace_bvh_perray" | grep "1048576" > $OUT/pair$MODE.pmc.txt
done
unset NTR_TRACE_PAIR
python3 - $OUT <<'PY'
import re, sys
out = sys.argv[1]
for mode in (0, 1):
    d = {}
    for l in open("%s/pair%d.pmc.txt" % (out, mode)):
        m = re.match(r"^(.*?)\s+grid=\s*(\d+)\s+(\S+)\s+n=\s*\d+\s+mean=(\S+)$", l.rstrip())
        if m:
            d.setdefault(m.group(1).strip(), {})[m.group(3)] = float(m.group(4))
    for k, c in d.items():
        cyc = c.get("SQ_INSTS_VALU@cycles") or 1.0
        print("pair=%d %-44s waves %6.0f  launch %.1f us  VALU/wave %6.0f  VMEM_RD/wave %5.1f  valu_issue %.3f  ta_busy %.3f  wait_share %.3f  clock %.2f GHz" % (
            mode, k, c.get("SQ_WAVES", 0), c.get("SQ_INSTS_VALU@ns", 0) * 1e-3, c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_WAVES", 1), 1),
            c.get("SQ_INSTS_VMEM_RD", 0) / max(c.get("SQ_WAVES", 1), 1), c.get("SQ_INSTS_VALU", 0) * 2 / (1024 * cyc),
            c.get("TA_TA_BUSY_sum", 0) / 256 / (c.get("TA_TA_BUSY_sum@cycles") or 1.0), c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1),
            cyc / max(c.get("SQ_INSTS_VALU@ns", 1), 1)))
PY
