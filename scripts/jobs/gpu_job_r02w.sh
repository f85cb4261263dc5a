#!/bin/bash
set -u
OUT=gpurun_out/r02w; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python3 scripts/ray_sort_study.py > $OUT/ray_sort_study.jsonl 2> $OUT/ray_sort_study.err; echo "rc=$?"; cat $OUT/ray_sort_study.jsonl; tail -n 3 $OUT/ray_sort_study.err
