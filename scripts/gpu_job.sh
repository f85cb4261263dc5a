#!/bin/bash
# One parameterised runner for GPU-box jobs (replaces the one-off scripts/jobs/gpu_job_r02*.sh of round 2):
#   gpurun --timeout 1500 -- 'bash scripts/gpu_job.sh <tag> <step> [<step> ...]'
# Every step writes under gpurun_out/<tag>/ and is bounded by its own `timeout`.  Steps:
#   tests            python -m pytest tests -m gpu -x -q
#   tests:<expr>     ... with -k <expr>
#   bench            python3 bench.py (default arguments) -> bench.json
#   bench_quick      bench.py --no-extras --no-cpu-baseline
#   bench_dist1      the RCCL path at world size 1 (NTR_BENCH_FORCE_DIST=1)
#   matrix:<scenes>  scripts/kernel_matrix.py <scenes> (comma separated)
#   configs          scripts/config_table.py
#   profile[:what]   scripts/profile_round.sh <tag> [lbvh|trace|hbm ...] (':' separated list)
#   prof_bench       rocprofv3 --kernel-trace --stats of the bench command itself
#   shard            scripts/studies/shard_balance_study.py
#   lbvh             scripts/studies/lbvh_sweep3.py
#   fuzz:<seconds>   tests/fuzz_parity.py for that long
#   ab:<reps>        scripts/ab_bench.sh: bench.py alternately against libntrace_amd.so and a second build (AB_LIB, default ntrace_amd/libntrace_amd_b.so)
#   knob:<VAR>:<v1>:<v2>...   scripts/studies/bench_knob.sh: the bench step under each value of one run-time tunable ("-" = unset), interleaved
#   py:<script>[:args...]   python3 scripts/<script>.py (or scripts/studies/<script>.py) args (':' separated); pyexp: the same with a build that has scripts/studies/rejected_patches/experiment_builds_trace.patch applied (ntrace_amd/libntrace_amd_diag.so)
set -u
TAG=${1:?tag}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for STEP in "$@"; do
  NAME=${STEP%%:*}; ARG=""; [[ "$STEP" == *:* ]] && ARG=${STEP#*:}
  echo "=== $STEP"
  case $NAME in
  tests)
    if [ -n "$ARG" ]; then timeout -k 5 1500 python3 -m pytest tests -m gpu -x -q -k "$ARG" > $OUT/tests.log 2>&1
    else timeout -k 5 1500 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; fi
    echo "rc=$?"; tail -n 15 $OUT/tests.log ;;
  bench)
    timeout -k 5 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"; tail -c 3000 $OUT/bench.json; tail -n 3 $OUT/bench.err ;;
  bench_quick)
    timeout -k 5 600 python3 bench.py --no-extras --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "rc=$?"; tail -c 1500 $OUT/bench_quick.json ;;
  bench_dist1)
    NTR_BENCH_FORCE_DIST=1 timeout -k 5 600 python3 bench.py --no-extras --no-cpu-baseline > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err; echo "rc=$?"; tail -c 1200 $OUT/bench_dist1.json ;;
  matrix)
    timeout -k 5 1200 python3 scripts/kernel_matrix.py $ARG 4 > $OUT/kernel_matrix_${ARG//,/_}.jsonl 2> $OUT/kernel_matrix.err; echo "rc=$?"
    python3 - $OUT/kernel_matrix_${ARG//,/_}.jsonl <<'EOF'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print("%-10s %-10s %-32s %-34s %8.3f ms %8.0f Mrays/s eq=%s" % (d["scene"], d["batch"], d["kernel"], d["env"], d["ms_min"], d["mrays"], d["records_equal_perray"]))
EOF
    tail -n 3 $OUT/kernel_matrix.err ;;
  configs)
    timeout -k 5 1500 python3 scripts/config_table.py $ARG > $OUT/config_table.md 2> $OUT/config_table.err; echo "rc=$?"; cat $OUT/config_table.md; tail -n 3 $OUT/config_table.err ;;
  profile)
    timeout -k 5 2400 bash scripts/profile_round.sh $TAG ${ARG//:/ } > $OUT/profile.log 2>&1; echo "rc=$?"; tail -n 5 $OUT/profile.log ;;
  prof_bench)
    timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 bench.py --no-extras --no-cpu-baseline > $OUT/prof_bench.log 2>&1; echo "rc=$?"
    python3 scripts/summarize_rocprof.py trace $OUT/prof_bench > $OUT/bench_kernel_summary.txt 2>&1; head -n 12 $OUT/bench_kernel_summary.txt
    find $OUT/prof_bench -name "*.csv" -size +8M -delete ;;
  shard)
    timeout -k 5 1200 python3 scripts/studies/shard_balance_study.py ${ARG//:/ } > $OUT/shard_balance.jsonl 2> $OUT/shard_balance.err; echo "rc=$?"; cat $OUT/shard_balance.jsonl; tail -n 3 $OUT/shard_balance.err ;;
  lbvh)
    timeout -k 5 900 python3 scripts/studies/lbvh_sweep3.py ${ARG//:/ } > $OUT/lbvh_sweep.jsonl 2> $OUT/lbvh_sweep.err; echo "rc=$?"; cat $OUT/lbvh_sweep.jsonl; tail -n 3 $OUT/lbvh_sweep.err ;;
  fuzz)
    timeout -k 5 $((ARG + 120)) python3 tests/fuzz_parity.py --seconds $ARG --seed ${FUZZ_SEED:-41} --progress $OUT/fuzz_progress.json > $OUT/fuzz.json 2> $OUT/fuzz.err; echo "rc=$?"; tail -c 1500 $OUT/fuzz.json; tail -n 3 $OUT/fuzz.err ;;
  ab)      # interleaved A/B of libntrace_amd.so and a second build of the library: ab:<reps>
    AB_OUT=$OUT bash scripts/ab_bench.sh ${ARG:-3} 2>&1 | tail -n 4 ;;
  knob)
    VAR=${ARG%%:*}; VALS=${ARG#*:}
    timeout -k 5 1500 bash scripts/studies/bench_knob.sh $OUT/knob_$VAR.jsonl $VAR ${VALS//:/ } > $OUT/knob_$VAR.txt 2>&1; echo "rc=$?"; cat $OUT/knob_$VAR.txt ;;
  sh)      # sh:<script>[:args...]  bash scripts/studies/<script>.sh args
    SCRIPT=${ARG%%:*}; REST=""; [[ "$ARG" == *:* ]] && REST=${ARG#*:}
    timeout -k 5 1500 bash scripts/studies/$SCRIPT.sh ${REST//:/ } > $OUT/$SCRIPT.out 2>&1; echo "rc=$?"; tail -n 30 $OUT/$SCRIPT.out ;;
  pyexp)   # py: with a diagnostic build of the library (per-wave timeline hooks: a patch, see scripts/studies/INDEX.md)
    SCRIPT=${ARG%%:*}; REST=""; [[ "$ARG" == *:* ]] && REST=${ARG#*:}
    [ -f scripts/$SCRIPT.py ] || SCRIPT=studies/$SCRIPT; mkdir -p $OUT/studies
    NTR_LIB_OVERRIDE=ntrace_amd/libntrace_amd_diag.so timeout -k 5 1500 python3 scripts/$SCRIPT.py ${REST//:/ } > $OUT/$SCRIPT.out 2> $OUT/$SCRIPT.err; echo "rc=$?"; tail -n 40 $OUT/$SCRIPT.out | cut -c1-1500; tail -n 5 $OUT/$SCRIPT.err ;;
  py)
    SCRIPT=${ARG%%:*}; REST=""; [[ "$ARG" == *:* ]] && REST=${ARG#*:}
    [ -f scripts/$SCRIPT.py ] || SCRIPT=studies/$SCRIPT; mkdir -p $OUT/studies
    timeout -k 5 1500 python3 scripts/$SCRIPT.py ${REST//:/ } > $OUT/$SCRIPT.out 2> $OUT/$SCRIPT.err; echo "rc=$?"; tail -n 40 $OUT/$SCRIPT.out; tail -n 5 $OUT/$SCRIPT.err ;;
  *) echo "unknown step $STEP" ;;
  esac
done
