#!/bin/bash
# Collector benches of a round: one-launch (fw_collect_step) against the three-launch collector per task / env count, and the
# in-launch timeline of fw_collect_step.  Run on the GPU box from the repo root; summaries land in gpurun_out/${ROUND}_profiles/.
ROUND=${ROUND:-r04}
OUT=gpurun_out/${ROUND}_profiles
mkdir -p "$OUT"
rm -f "$OUT/${ROUND}_rollout_bench.jsonl" "$OUT/${ROUND}_rollout_bench_three_launch.jsonl"
for spec in "waypoints 4096" "waypoints 2048" "waypoints 8192" "objlock 4096" "combined 4096" "combined 2048"; do
  timeout -k 10 300 python tools/bench_rollout.py $spec >> "$OUT/${ROUND}_rollout_bench.jsonl" 2>> "$OUT/rollout_bench.err" || exit 1
  timeout -k 10 300 python tools/bench_rollout.py $spec three >> "$OUT/${ROUND}_rollout_bench_three_launch.jsonl" 2>> "$OUT/rollout_bench.err" || exit 1
done
timeout -k 10 200 python tools/trace_collect.py waypoints 4096 > "$OUT/${ROUND}_collect_step_trace.txt" 2>> "$OUT/rollout_bench.err" || exit 1
if [ -f tools/_build/libfwsim_ppoprof.so ]; then timeout -k 10 300 python tools/prof_ppo.py > "$OUT/${ROUND}_ppo_update_cycles.txt" 2>> "$OUT/rollout_bench.err" || exit 1; cat "$OUT/${ROUND}_ppo_update_cycles.txt"; fi
cut -c1-170 "$OUT/${ROUND}_rollout_bench.jsonl" "$OUT/${ROUND}_rollout_bench_three_launch.jsonl"; cat "$OUT/${ROUND}_collect_step_trace.txt"
