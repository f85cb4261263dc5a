#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command itself (the roofline's launch duration must agree with it)
set -u
OUT=gpurun_out/r02y; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 bench.py --no-extras --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"
{ echo "== rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extras --no-cpu-baseline"; tail -n 1 $OUT/bench.json; python3 scripts/summarize_rocprof.py trace $OUT/bench_trace; } > $OUT/bench_kernel_summary.txt 2>&1
head -c 1500 $OUT/bench_kernel_summary.txt | cut -c1-200; echo; sed -n 3,14p $OUT/bench_kernel_summary.txt | cut -c1-150
find $OUT -name "*_kernel_trace.csv" -size +8M -delete
