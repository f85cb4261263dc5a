#!/bin/bash
# GPU job r02v: final round-2 state: full GPU suite, fuzz, bench line, config table with the 10 M-triangle host SAH build
set -u
OUT=gpurun_out/r02v; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1500 python3 -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -n 5 $OUT/pytest_gpu.log
timeout -k 5 500 python3 tests/fuzz_parity.py --seconds 240 --seed 34 > $OUT/fuzz34.json 2> $OUT/fuzz34.err; echo "fuzz rc=$?"; tail -c 600 $OUT/fuzz34.json
timeout -k 5 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 300 $OUT/bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r02v/bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'primary',d['primary_mrays'],'ao',d['ao_mrays'],'binding',d['roofline'].get('binding'))
print('lbvh',d['extras']['lbvh']['build_ms'],d['extras']['lbvh']['roofline'])
h=d['extras']['hbm_resident_point']; print('hbm lbvh',h['lbvh_build']['build_ms'],h['lbvh_build'].get('roofline')); print('hbm primary',h['primary']['ms'],'incoherent',h['incoherent']['ms'])
print('overlapped',d['extras']['overlapped_frame'],'cpu',d['cpu_baseline']['value'],d['cpu_baseline']['parity_mismatches_whole_step'])
PY
NTR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 5 400 python3 bench.py --no-extras --no-cpu-baseline > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err; echo "bench dist rc=$?"; tail -c 400 $OUT/bench_dist1.json
timeout -k 5 1500 python3 scripts/config_table.py --sah-10m > $OUT/config_table.txt 2> $OUT/config_table.err; echo "table rc=$?"; cat $OUT/config_table.txt
