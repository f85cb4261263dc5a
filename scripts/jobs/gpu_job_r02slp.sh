#!/bin/bash
# the whole library built with -fno-slp-vectorize: GPU suite, LBVH sweep, bench rounds
set -u
OUT=gpurun_out/r02slp; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1500 python3 -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 2 $OUT/pytest_gpu.log
timeout 300 python3 scripts/lbvh_sweep3.py 2>/dev/null | grep '"cfg": {}' | cut -c1-260
for R in 1 2; do
  timeout -k 5 300 python3 bench.py --no-extras --no-cpu-baseline --steps 40 > $OUT/b_$R.json 2> $OUT/b.err
  python3 - <<PY
import json
d=json.loads(open('$OUT/b_$R.json').read().strip().splitlines()[-1])
print('round $R value %.0f primary %.0f ao %.0f primary_ms %.4f' % (d['value'], d['primary_mrays'], d['ao_mrays'], d['kernel_ms']['primary']))
PY
done
