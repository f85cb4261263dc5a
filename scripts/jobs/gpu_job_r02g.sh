#!/bin/bash
set -u
OUT=gpurun_out/r02g; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python3 -m pytest tests/test_rayops_gpu.py tests/test_lbvh_gpu.py tests/test_bench_obj_gpu.py -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 6 $OUT/pytest.log
timeout -k 5 600 python3 scripts/ray_sort_study.py > $OUT/ray_sort_study.jsonl 2> $OUT/ray_sort_study.err; echo "study rc=$?"; cat $OUT/ray_sort_study.jsonl; tail -n 3 $OUT/ray_sort_study.err
