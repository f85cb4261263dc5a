#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 300 python3 scripts/lbvh_sweep3.py 2>/dev/null | grep '"cfg": {}' | cut -c1-260
timeout 300 python3 -m pytest tests/test_lbvh_gpu.py -m gpu -q -x -k "default" 2>&1 | tail -2
