#!/bin/bash
# bench.py protocol numbers for the four combinations of the scalar node fetch and the octant slabs (two rounds each, interleaved)
set -u
OUT=gpurun_out/r02s; mkdir -p $OUT
export TMPDIR=/tmp
for R in 1 2 3; do
for U in 0; do for O in 0 1; do
  NTR_TRACE_UNIFORM=$U NTR_TRACE_OCTANT=$O timeout -k 5 300 python3 bench.py --no-extras --no-cpu-baseline --steps 40 > $OUT/b_${U}_${O}_$R.json 2> $OUT/b.err
  python3 - <<PY
import json
d=json.loads(open('$OUT/b_${U}_${O}_$R.json').read().strip().splitlines()[-1])
print('round $R uniform=$U octant=$O value %.0f primary %.0f ao %.0f primary_ms %.4f' % (d['value'], d['primary_mrays'], d['ao_mrays'], d['kernel_ms']['primary']))
PY
done; done; done
