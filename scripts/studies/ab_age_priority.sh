mkdir -p gpurun_out/r04d; export TMPDIR=/tmp
for LIB in libntrace_amd.so libntrace_amd_ab_age6.so libntrace_amd_ab_age8.so; do
  for WL in "hairball diffuse" "hairball incoherent" "courtyard incoherent" "courtyard diffuse"; do
    NTR_TRACE_HANDOFF=0 NTR_LIB_OVERRIDE=ntrace_amd/$LIB timeout 200 python3 scripts/studies/handoff_workload.py $WL 7 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$LIB', d['scene'], d['batch'], 'min %.3f'%min(d['ms'][2:]), d['ms'])"
  done
done
