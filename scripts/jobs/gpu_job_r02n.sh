#!/bin/bash
set -u
OUT=gpurun_out/r02n; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python3 scripts/lbvh_sweep3.py > $OUT/lbvh_sweep.jsonl 2> $OUT/lbvh_sweep.err; echo "sweep rc=$?"; grep -v "LEGACY\|EMIT\|AGG" $OUT/lbvh_sweep.jsonl | cut -c1-330
