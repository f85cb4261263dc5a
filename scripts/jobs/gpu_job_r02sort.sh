#!/bin/bash
set -u
export TMPDIR=/tmp
for I in 16 24 32; do
  echo "items $I"; NTR_LBVH_SORT_ITEMS=$I timeout 300 python3 scripts/lbvh_sweep3.py courtyard hairball 2>/dev/null | grep '"cfg": {}' | cut -c1-260
done
