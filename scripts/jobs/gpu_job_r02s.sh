#!/bin/bash
# bench.py protocol numbers (three rounds) + trace parity tests
set -u
OUT=gpurun_out/r02s; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_trace_gpu.py tests/test_kat_gpu.py -m gpu -q -x --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 2 $OUT/pytest.log
for R in 1 2 3; do
  timeout -k 5 300 python3 bench.py --no-extras --no-cpu-baseline --steps 40 > $OUT/b_$R.json 2> $OUT/b.err
  python3 - <<PY
import json
d=json.loads(open('$OUT/b_$R.json').read().strip().splitlines()[-1])
print('round $R value %.0f primary %.0f ao %.0f primary_ms %.4f' % (d['value'], d['primary_mrays'], d['ao_mrays'], d['kernel_ms']['primary']))
PY
done
