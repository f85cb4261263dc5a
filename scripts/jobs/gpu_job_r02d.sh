#!/bin/bash
# GPU job r02d: per-kernel profile of the new LBVH pipeline (kernel trace only) + lbvh tests on the changed scan / sort look-back.
set -u
OUT=gpurun_out/r02d; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python3 -m pytest tests/test_lbvh_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x --timeout 600 > $OUT/pytest_lbvh.log 2>&1; echo "pytest lbvh rc=$?"; tail -n 4 $OUT/pytest_lbvh.log
for S in atrium hairball courtyard; do
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lbvh_$S -- python3 scripts/workloads.py lbvh $S 8 > $OUT/lbvh_$S.log 2>&1
  { tail -n 1 $OUT/lbvh_$S.log | cut -c1-600; python3 scripts/summarize_rocprof.py trace $OUT/lbvh_$S; } > $OUT/lbvh_$S.kernels.txt 2>&1
  head -n 16 $OUT/lbvh_$S.kernels.txt | cut -c1-170
done
timeout -k 5 300 python3 scripts/lbvh_sweep3.py > $OUT/lbvh_sweep.jsonl 2> $OUT/lbvh_sweep.err
find $OUT -name "*.csv" -size +4M -delete
