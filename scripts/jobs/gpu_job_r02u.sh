#!/bin/bash
set -u
OUT=gpurun_out/r02u; mkdir -p $OUT
export TMPDIR=/tmp
for S in atrium courtyard; do
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$S -- python3 scripts/workloads.py lbvh $S 8 > $OUT/$S.log 2>&1
python3 scripts/summarize_rocprof.py trace $OUT/$S | head -14 | cut -c1-150
done
