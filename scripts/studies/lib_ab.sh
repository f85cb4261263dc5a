#!/bin/bash
# lib_ab.sh <out> <scenes> <libA.so> <libB.so> [reps]: alternate two builds of the library on one box (scripts/studies/lib_ab.py)
OUT=$1; SC=$2; A=$3; B=$4; REPS=${5:-2}
mkdir -p $(dirname $OUT); : > $OUT
for i in $(seq $REPS); do
  for L in $A $B; do NTR_LIB_OVERRIDE=ntrace_amd/$L timeout 600 python3 scripts/studies/lib_ab.py $SC 2>/dev/null | tail -n 1 >> $OUT; done
done
python3 - $OUT <<'PY'
import json, sys, collections
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    for k, v in r["ms"].items(): agg[k][r["lib"]].append(v)
libs = sorted({r["lib"] for r in rows})
for k, d in agg.items():
    print("%-44s " % k + "  ".join("%s min %.4f" % (l.split("/")[-1], min(d[l])) for l in libs) + ("   B/A %.3f" % (min(d[libs[1]]) / min(d[libs[0]])) if len(libs) == 2 else ""))
PY
