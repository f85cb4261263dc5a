#!/bin/bash
set -u
OUT=gpurun_out/r02t; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_lbvh_gpu.py -m gpu -q -x --timeout 600 > $OUT/pytest_lbvh.log 2>&1; echo "pytest lbvh rc=$?"; tail -n 15 $OUT/pytest_lbvh.log
