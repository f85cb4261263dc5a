#!/bin/bash
# final state of round 2: smoke(), the full GPU suite, a default bench run
set -u
OUT=gpurun_out/r02final; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $OUT/smoke.log
timeout -k 5 1500 python3 -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest_gpu.log
timeout -k 5 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 200 $OUT/bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r02final/bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'primary',d['primary_mrays'],'ao',d['ao_mrays'],'host_sah_s',d['host_sah_build_s'])
print('binding',d['roofline'].get('binding'))
print('lbvh',d['extras']['lbvh']['build_ms'],'hbm lbvh',d['extras']['hbm_resident_point']['lbvh_build']['build_ms'])
PY
