#!/bin/bash
# Usage: scripts/pmc.sh <tag> "<counter set 1>" "<counter set 2>" ... -- bench args
# One rocprofv3 --pmc pass per counter set (never combined with other trace domains than kernel-trace).
TAG=$1; shift
SETS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done
shift || true
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
i=0
for C in "${SETS[@]}"; do
  timeout -k 5 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/set$i -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $OUT/set$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/set*/")):
    for f in glob.glob(d+"/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "trace_bvh" in k:
                agg[k.split("(")[0][-40:] + " grid=" + r.get("Grid_Size","?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            for c, v in cs.items():
                print("%-60s %-36s n=%3d mean=%.4g" % (k, c, len(v), sum(v)/len(v)))
PY
