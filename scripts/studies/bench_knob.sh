#!/bin/bash
# bench_knob.sh <out.jsonl> <NTR_VARIABLE> <value> [<value> ...]: the bench step (bench.py --no-extras --no-cpu-baseline) under each value of one
# run-time tunable, interleaved twice on one box ("-" = variable unset; BENCH_ARGS = further bench.py arguments, e.g. --kernel); prints primary /
# AO launch times and the value per setting.
OUT=$1; VAR=$2; shift 2
mkdir -p $(dirname $OUT); : > $OUT
for rep in 1 2; do
  for V in "$@"; do
    if [ "$V" = "-" ]; then unset $VAR; else export $VAR=$V; fi
    timeout 300 python3 bench.py --no-extras --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -n 1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print(json.dumps(dict(var='$VAR', value='$V', mrays=d['value'], primary_ms=d['kernel_ms']['primary'], ao_ms=d['kernel_ms']['ao_total'], cold=d['cold_dispatch_order']['mrays'] if d.get('cold_dispatch_order') else None)))" >> $OUT
  done
done
unset $VAR
python3 - $OUT <<'PY'
import json, sys, collections
agg = collections.OrderedDict()
for l in open(sys.argv[1]):
    r = json.loads(l); agg.setdefault(r["value"], []).append(r)
for v, rs in agg.items():
    print("%s=%-6s  value %8.0f  primary %.4f ms  ao %.4f ms  cold %s" % (rs[0]["var"], v, max(r["mrays"] for r in rs), min(r["primary_ms"] for r in rs), min(r["ao_ms"] for r in rs), max((r["cold"] or 0) for r in rs)))
PY
