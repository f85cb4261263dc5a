#!/bin/bash
set -u
OUT=gpurun_out/r02flags; mkdir -p $OUT
export TMPDIR=/tmp
for V in base v1 v2 v3; do
  if [ $V = base ]; then unset NTR_LIB_OVERRIDE; else export NTR_LIB_OVERRIDE=$PWD/ntrace_amd/libntrace_amd_$V.so; fi
  for R in 1 2; do
  timeout -k 5 300 python3 bench.py --no-extras --no-cpu-baseline --steps 40 > $OUT/b_${V}_$R.json 2> $OUT/b.err
  python3 - <<PY
import json
d=json.loads(open('$OUT/b_${V}_$R.json').read().strip().splitlines()[-1])
print('$V round $R value %.0f primary %.0f ao %.0f' % (d['value'], d['primary_mrays'], d['ao_mrays']))
PY
  done
done
