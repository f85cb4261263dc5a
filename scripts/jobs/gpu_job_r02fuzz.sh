#!/bin/bash
set -u
OUT=gpurun_out/r02fuzz; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 400 python3 tests/fuzz_parity.py --seconds 200 --seed 35 > $OUT/fuzz35.json 2> $OUT/fuzz35.err; echo "fuzz rc=$?"; tail -c 700 $OUT/fuzz35.json
