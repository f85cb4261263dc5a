#!/bin/bash
# GPU job r02k: persistent kernels with 64 pool heads + static first chunks: parity tests, timelines against the per-ray kernel.
set -u
OUT=gpurun_out/r02k; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python3 -m pytest tests/test_trace_gpu.py tests/test_kat_gpu.py tests/test_host_cpp.py -m gpu -q -x --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -n 5 $OUT/pytest_gpu.log
NTR_LIB_OVERRIDE=$PWD/ntrace_amd/libntrace_amd_exp.so timeout -k 5 400 python3 scripts/persist_diag.py > $OUT/persist_diag.jsonl 2> $OUT/persist_diag.err; echo "diag rc=$?"
python3 - <<'PY'
import json
for l in open('gpurun_out/r02k/persist_diag.jsonl'):
    d=json.loads(l)
    print(d['kernel'][:8],d['label'],d.get('tunables'),'us',round(d['us_plain']),'life_mean',round(d['life_us']['mean']),'first_end',round(d['first_end_us']),d.get('refill'))
PY
