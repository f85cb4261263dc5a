#!/bin/bash
set -u
export TMPDIR=/tmp
NTR_LIB_OVERRIDE=$PWD/ntrace_amd/libntrace_amd_exp.so timeout 600 python3 scripts/ao_cost_correlation.py 2>&1 | tail -6
