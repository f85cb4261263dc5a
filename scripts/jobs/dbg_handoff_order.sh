# debug: is tests/test_handoff_gpu.py order dependent / flaky in the full GPU tier?
for i in 1 2 3; do
  echo "== full run $i"; timeout 900 python3 -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -n 2
done
echo "== handoff first"; timeout 900 python3 -m pytest tests/test_handoff_gpu.py tests/test_bench_obj_gpu.py tests/test_configs_gpu.py tests/test_dist_native_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -n 2
