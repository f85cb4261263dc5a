#!/bin/bash
set -u
export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_trace_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -3
