#!/bin/bash
set -u
OUT=gpurun_out/r02q; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_lbvh_gpu.py -m gpu -q -x --timeout 600 > $OUT/pytest_lbvh.log 2>&1; echo "pytest lbvh rc=$?"; tail -n 3 $OUT/pytest_lbvh.log
timeout -k 5 300 python3 tests/fuzz_parity.py --seconds 60 --seed 33 > $OUT/fuzz33.json 2> $OUT/fuzz33.err; echo "fuzz rc=$?"; tail -c 400 $OUT/fuzz33.json
timeout -k 5 600 python3 scripts/lbvh_sweep3.py > $OUT/lbvh_sweep.jsonl 2> $OUT/lbvh_sweep.err; echo "sweep rc=$?"; grep -v "LEGACY\|EMIT\|AGG\|MARK" $OUT/lbvh_sweep.jsonl | cut -c1-330
timeout -k 5 1500 python3 scripts/config_table.py > $OUT/config_table.txt 2> $OUT/config_table.err; echo "table rc=$?"; cat $OUT/config_table.txt; tail -n 3 $OUT/config_table.err
