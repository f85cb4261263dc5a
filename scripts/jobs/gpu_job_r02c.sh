#!/bin/bash
# GPU job r02c: bottom-up LBVH emit (tests + fuzz + sweep), ageing priority A/B, TA counters.
set -u
OUT=gpurun_out/r02c; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_lbvh_gpu.py tests/test_configs_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x --timeout 600 > $OUT/pytest_lbvh.log 2>&1; echo "pytest lbvh rc=$?"
tail -n 25 $OUT/pytest_lbvh.log
timeout -k 5 300 python3 tests/fuzz_parity.py --seconds 120 --seed 21 > $OUT/fuzz21.json 2> $OUT/fuzz21.err; echo "fuzz rc=$?"; tail -c 400 $OUT/fuzz21.json
timeout -k 5 600 python3 scripts/lbvh_sweep3.py > $OUT/lbvh_sweep.jsonl 2> $OUT/lbvh_sweep.err; echo "sweep rc=$?"
timeout -k 5 300 python3 scripts/age_prio_ab.py > $OUT/age_prio.jsonl 2> $OUT/age_prio.err; echo "age rc=$?"
for C in "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" "TA_BUFFER_WAVEFRONTS_sum"; do
  N=$(echo $C | tr ' ' '+')
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 scripts/workloads.py trace atrium fermi_speculative_while_while 4 > $OUT/pmc_$N.log 2>&1
done
python3 scripts/summarize_rocprof.py pmc $OUT/pmc_* > $OUT/ta_pmc.txt 2>&1; grep -E "perray<4, false" $OUT/ta_pmc.txt | head
find $OUT -name "*.csv" -size +4M -delete
