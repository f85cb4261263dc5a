#!/bin/bash
set -u
OUT=gpurun_out/r02pd; mkdir -p $OUT
export TMPDIR=/tmp
for R in 1 2; do for D in 0 7 8 9 10; do
  if [ $D = 0 ]; then export NTR_TRACE_PREDICT=0; else export NTR_TRACE_PREDICT=1 NTR_TRACE_PREDICT_DEPTH=$D; fi
  timeout -k 5 300 python3 bench.py --no-extras --no-cpu-baseline --steps 40 > $OUT/b_${D}_$R.json 2> $OUT/b.err
  python3 - <<PY
import json
d=json.loads(open('$OUT/b_${D}_$R.json').read().strip().splitlines()[-1])
print('round $R predict_depth=$D value %.0f primary %.0f ao %.0f' % (d['value'], d['primary_mrays'], d['ao_mrays']))
PY
done; done
