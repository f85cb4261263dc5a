#!/bin/bash
# GPU job r02e: full GPU suite + longer fuzz on the bottom-up LBVH, then bench.
set -u
OUT=gpurun_out/r02e; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1500 python3 -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 4 $OUT/pytest_gpu.log
timeout -k 5 400 python3 tests/fuzz_parity.py --seconds 240 --seed 22 > $OUT/fuzz22.json 2> $OUT/fuzz22.err; echo "fuzz rc=$?"; tail -c 500 $OUT/fuzz22.json
timeout -k 5 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 300 $OUT/bench.err
