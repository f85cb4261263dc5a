#!/bin/bash
# GPU job r02l: LBVH bottom-up emit writing final nodes / Woop rows directly (no records, no finalize pass): parity on all build paths, fuzz, timings.
set -u
OUT=gpurun_out/r02l; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_lbvh_gpu.py -m gpu -q -x --timeout 600 > $OUT/pytest_lbvh.log 2>&1; echo "pytest lbvh rc=$?" | tee -a $OUT/pytest_lbvh.log
tail -n 15 $OUT/pytest_lbvh.log
timeout -k 5 300 python3 tests/fuzz_parity.py --seconds 90 --seed 31 > $OUT/fuzz31.json 2> $OUT/fuzz31.err; echo "fuzz rc=$?"; tail -c 700 $OUT/fuzz31.json
timeout -k 5 600 python3 scripts/lbvh_sweep3.py > $OUT/lbvh_sweep.jsonl 2> $OUT/lbvh_sweep.err; echo "sweep rc=$?"; tail -n 12 $OUT/lbvh_sweep.jsonl
