#!/bin/bash
set -u
OUT=gpurun_out/r02z; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 5 900 python3 scripts/shard_balance_study.py > $OUT/shard_balance.jsonl 2> $OUT/shard_balance.err; echo "rc=$?"; cat $OUT/shard_balance.jsonl; tail -n 3 $OUT/shard_balance.err
