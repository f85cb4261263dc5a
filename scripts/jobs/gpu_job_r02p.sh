#!/bin/bash
set -u
OUT=gpurun_out/r02p; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_trace_gpu.py tests/test_kat_gpu.py -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest.log
timeout -k 5 600 python3 scripts/uniform_ab.py fermi_speculative_while_while > $OUT/uniform_ab.jsonl 2> $OUT/uniform_ab.err; echo "ab rc=$?"; cat $OUT/uniform_ab.jsonl; tail -n 3 $OUT/uniform_ab.err
