#!/bin/bash
set -u
OUT=gpurun_out/r02x; mkdir -p $OUT
export TMPDIR=/tmp
NTR_SAH_TIMING=1 timeout -k 5 1500 python3 scripts/config_table.py --sah-10m > $OUT/config_table.txt 2> $OUT/config_table.err; echo "table rc=$?"; cat $OUT/config_table.txt; grep "SAHBVHBuilder\|ntr_sah_build" $OUT/config_table.err | tail -4
timeout -k 5 300 python3 -m pytest tests/test_configs_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -2
