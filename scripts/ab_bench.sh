#!/bin/bash
# Interleaved A/B of two builds of the library on ONE GPU box (boxes differ by +-1.5 %, so only same-box alternation ranks variants):
#   scripts/ab_bench.sh <reps> [bench.py args...]      A = libntrace_amd.so, B = $AB_LIB (default ntrace_amd/libntrace_amd_b.so: build the
#   variant, copy the library there, rebuild the product)
# Prints value / primary / AO Mrays/s per run and the means.
REPS=${1:-3}; shift || true
OUT=${AB_OUT:-gpurun_out/ab}; mkdir -p $OUT
: > $OUT/ab.jsonl
for i in $(seq 1 $REPS); do
  for V in A B; do
    LIB=ntrace_amd/libntrace_amd.so; [ $V = B ] && LIB=${AB_LIB:-ntrace_amd/libntrace_amd_b.so}
    NTR_LIB_OVERRIDE=$LIB timeout -k 5 300 python3 bench.py --no-extras --no-cpu-baseline "$@" 2> $OUT/ab_$V.err | tail -n 1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps(dict(variant='$V', value=d['value'], primary=d['primary_mrays'], ao=d['ao_mrays'], primary_ms=d['kernel_ms']['primary'], ao_ms=d['kernel_ms']['ao_total'])))" >> $OUT/ab.jsonl
  done
done
python3 - $OUT/ab.jsonl <<'PY'
import json, sys, collections
rows = collections.defaultdict(list)
for l in open(sys.argv[1]):
    d = json.loads(l); rows[d['variant']].append(d)
for v, r in sorted(rows.items()):
    m = lambda k: sum(x[k] for x in r) / len(r)
    print("%s n=%d value %.0f primary %.0f (%.4f ms) ao %.0f (%.4f ms)" % (v, len(r), m('value'), m('primary'), m('primary_ms'), m('ao'), m('ao_ms')))
if 'A' in rows and 'B' in rows:
    a = sum(x['value'] for x in rows['A']) / len(rows['A']); b = sum(x['value'] for x in rows['B']) / len(rows['B'])
    print("B / A = %.4f" % (b / a))
PY
