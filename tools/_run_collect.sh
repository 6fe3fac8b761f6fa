mkdir -p gpurun_out/r03e
timeout -k 10 500 python -m pytest tests/test_rollout_gpu.py -m gpu -x -q > gpurun_out/r03e/pytest.log 2>&1; tail -3 gpurun_out/r03e/pytest.log
timeout -k 10 200 python tools/trace_collect.py waypoints 4096 > gpurun_out/r03e/trace.txt 2>&1; cat gpurun_out/r03e/trace.txt
for spec in "waypoints 4096" "waypoints 8192" "waypoints 2048" "objlock 4096" "combined 4096" "combined 2048"; do
  for mode in one_launch three; do
    timeout -k 10 200 python tools/bench_rollout.py $spec $mode >> gpurun_out/r03e/bench.jsonl 2>> gpurun_out/r03e/bench.err
  done
done
cut -c1-175 gpurun_out/r03e/bench.jsonl
