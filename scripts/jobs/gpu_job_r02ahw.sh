#!/bin/bash
set -u
OUT=gpurun_out/r02ahw; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_trace_gpu.py tests/test_kat_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -2
for R in 1 2; do for W in 4 2 1; do
  NTR_TRACE_CLOSEST_WAVES=$W timeout -k 5 300 python3 bench.py --no-extras --no-cpu-baseline --steps 40 > $OUT/b_${W}_$R.json 2> $OUT/b.err
  python3 - <<PY
import json
d=json.loads(open('$OUT/b_${W}_$R.json').read().strip().splitlines()[-1])
print('round $R closest_waves=$W value %.0f primary %.0f ao %.0f' % (d['value'], d['primary_mrays'], d['ao_mrays']))
PY
done; done
NTR_TRACE_CLOSEST_WAVES=1 timeout -k 5 900 python3 -m pytest tests/test_trace_gpu.py tests/test_kat_gpu.py tests/test_configs_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -2
