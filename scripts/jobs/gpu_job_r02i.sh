#!/bin/bash
# Full GPU suite, ray-sort study, bench (default) -- state before the final profiles.
set -u
OUT=gpurun_out/r02i; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1500 python3 -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 4 $OUT/pytest_gpu.log
timeout -k 5 600 python3 scripts/ray_sort_study.py > $OUT/ray_sort_study.jsonl 2> $OUT/ray_sort_study.err; echo "study rc=$?"; cat $OUT/ray_sort_study.jsonl
timeout -k 5 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 300 $OUT/bench.err
