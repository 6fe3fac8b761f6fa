#!/bin/bash
# rocprofv3 kernel statistics of one collector / update bench run (tools/bench_rollout.py): per-kernel launch counts and
# durations of fw_collect_step's kernel, fw_gae, fw_ppo_update ...  Run on the GPU box from the repo root.
ROUND=${ROUND:-r04}; TASK=${1:-waypoints}; ENVS=${2:-4096}
OUT=$PWD/gpurun_out/${ROUND}_profiles; mkdir -p "$OUT"
export TMPDIR=/tmp
D=/tmp/prof_rollout_$$; rm -rf "$D"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -- python3 "$OLDPWD/tools/bench_rollout.py" "$TASK" "$ENVS" > "$OUT/${ROUND}_rollout_${TASK}_bench_under_profiler.json" 2> "$OUT/profile_rollout.err" ) || exit 1
f=$(find "$D" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/${ROUND}_rollout_${TASK}_kernel_stats.csv"
rm -rf "$D"
head -12 "$OUT/${ROUND}_rollout_${TASK}_kernel_stats.csv" | cut -c1-200
