#!/bin/bash
# ray splitting: launch times with / without, differing records, then the trace parity tests with splitting on
mkdir -p gpurun_out/split
[ -n "${DEBUG:-}" ] && for K in kepler_dynamic_fetch fermi_speculative_while_while; do python3 scripts/studies/split_debug.py $K ${DEBUG} 2>&1 | grep -v '^W2026\|amdgpu.ids' | tail -4 | cut -c1-300; done
for S in ${SCENES:-courtyard hairball}; do
  timeout 900 python3 scripts/studies/split_study.py $S ${SLICES:-0,8,16,32,0} ${KERNELS:-kepler_dynamic_fetch} ${BATCHES:-} 2>&1 | grep -v "^W2026\|amdgpu.ids" | tee -a gpurun_out/split/split_study_${TAG:-x}.jsonl
done
[ -n "${TESTS:-}" ] && NTR_TRACE_SPLIT_SLICE=${TEST_SLICE:-8} timeout 1500 python3 -m pytest tests -m gpu -x -q -k "$TESTS" 2>&1 | tail -n 8
true
