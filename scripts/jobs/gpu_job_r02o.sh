#!/bin/bash
# GPU job r02o: full GPU suite, fuzz, bench line after the LBVH direct-write pipeline and the persistent-kernel pool heads.
set -u
OUT=gpurun_out/r02o; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1500 python3 -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -n 5 $OUT/pytest_gpu.log
timeout -k 5 400 python3 tests/fuzz_parity.py --seconds 150 --seed 32 > $OUT/fuzz32.json 2> $OUT/fuzz32.err; echo "fuzz rc=$?"; tail -c 600 $OUT/fuzz32.json
timeout -k 5 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 300 $OUT/bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r02o/bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'primary',d['primary_mrays'],'ao',d['ao_mrays'],'binding',d['roofline'].get('binding'))
print('lbvh',d['extras']['lbvh']['build_ms'],d['extras']['lbvh']['phases_ms'],d['extras']['lbvh']['roofline'])
h=d['extras']['hbm_resident_point']; print('hbm lbvh',h['lbvh_build']['build_ms'],h['lbvh_build'].get('roofline'))
PY
