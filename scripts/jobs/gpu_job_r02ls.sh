#!/bin/bash
# bench-protocol sweep of the leaf-switch threshold with the final kernels
set -u
OUT=gpurun_out/r02ls; mkdir -p $OUT
export TMPDIR=/tmp
for R in 1 2; do for L in 16 24 32 48; do
  NTR_TRACE_LEAF_SWITCH=$L timeout -k 5 300 python3 bench.py --no-extras --no-cpu-baseline --steps 40 > $OUT/b_${L}_$R.json 2> $OUT/b.err
  python3 - <<PY
import json
d=json.loads(open('$OUT/b_${L}_$R.json').read().strip().splitlines()[-1])
print('round $R leaf_switch=$L value %.0f primary %.0f ao %.0f' % (d['value'], d['primary_mrays'], d['ao_mrays']))
PY
done; done
