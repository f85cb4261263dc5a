#!/bin/bash
# lib_abn.sh <out> <scenes> <reps> <lib.so> [<lib.so> ...]: alternate several builds of the library on one box (scripts/studies/lib_ab.py)
OUT=$1; SC=$2; REPS=$3; shift 3
mkdir -p $(dirname $OUT); : > $OUT
for i in $(seq $REPS); do
  for L in "$@"; do NTR_LIB_OVERRIDE=ntrace_amd/$L timeout 600 python3 scripts/studies/lib_ab.py $SC 2>/dev/null | tail -n 1 >> $OUT; done
done
python3 - $OUT <<'PY'
import json, sys, collections
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    for k, v in r["ms"].items(): agg[k][r["lib"]].append(v)
libs = sorted({r["lib"] for r in rows})
for k, d in agg.items():
    base = min(d[libs[0]])
    print("%-40s " % k + "  ".join("%s %.4f (%.3f)" % (l.split("/")[-1].replace("libntrace_amd", "").replace(".so", "") or "product", min(d[l]), min(d[l]) / base) for l in libs))
PY
