#!/bin/bash
# GPU job r02b: full GPU test suite on the new LBVH pipeline / C-ABI changes, persistent-kernel diagnosis, LBVH sweep, bench lines.
set -u
OUT=gpurun_out/r02b; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 1500 python3 -m pytest tests -m gpu -q -x --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -n 5 $OUT/pytest_gpu.log
NTR_LIB_OVERRIDE=$PWD/ntrace_amd/libntrace_amd_exp.so timeout -k 5 400 python3 scripts/persist_diag.py > $OUT/persist_diag.jsonl 2> $OUT/persist_diag.err; echo "diag rc=$?"
timeout -k 5 600 python3 scripts/lbvh_sweep2.py > $OUT/lbvh_sweep.jsonl 2> $OUT/lbvh_sweep.err; echo "sweep rc=$?"
timeout -k 5 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
NTR_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 5 400 python3 bench.py --no-extras --no-cpu-baseline > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err; echo "bench dist rc=$?"
tail -c 600 $OUT/bench.err
