#!/bin/bash
set -u
OUT=gpurun_out/r02xcd; mkdir -p $OUT
export TMPDIR=/tmp
NTR_TRACE_XCD=1 timeout -k 5 900 python3 -m pytest tests/test_trace_gpu.py tests/test_kat_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -2
for R in 1 2; do for X in 0 1; do
  NTR_TRACE_XCD=$X timeout -k 5 300 python3 bench.py --no-extras --no-cpu-baseline --steps 40 > $OUT/b_${X}_$R.json 2> $OUT/b.err
  python3 - <<PY
import json
d=json.loads(open('$OUT/b_${X}_$R.json').read().strip().splitlines()[-1])
print('round $R xcd=$X value %.0f primary %.0f ao %.0f' % (d['value'], d['primary_mrays'], d['ao_mrays']))
PY
done; done
