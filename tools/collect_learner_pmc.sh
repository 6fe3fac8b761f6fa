#!/bin/bash
# Counter evidence for the LEARNER kernels (fw_collect_kernel_*, fw_collect_close_kernel, fw_ppo_update_kernel, fw_render_kernel),
# on the GPU box through gpurun:    bash tools/collect_learner_pmc.sh [task] [envs]
# Passes (separate runs: FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never combined with a trace domain other than
# --kernel-trace; the program after `--` is python3 itself):
#   stats  rocprofv3 --kernel-trace --stats of the default (hipGraph-replayed) tools/bench_rollout.py run -> launch durations
#   fetch / write / sq / mfma / lds: counter passes of `tools/bench_rollout.py <task> <envs> pmc` (eager launches, 3 rollouts, 2 updates)
#   render: the same passes of `tools/bench_render.py <envs> 32 render_only`
# tools/summarize_learner_pmc.py turns gpurun_out/${ROUND}_learner/ into profiles/${ROUND}_learner_pmc.json.
set -o pipefail
export TMPDIR=/tmp
ROUND=${ROUND:-r04}
TASK=${1:-waypoints}; ENVS=${2:-4096}
OUT=$PWD/gpurun_out/${ROUND}_learner; mkdir -p "$OUT"
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1 || true
pick() { r=""; for c in "$@"; do grep -qw "$c" "$OUT/counters_list.txt" && r="$r $c"; done; echo $r; }
SQ=$(pick SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES)
MF=$(pick SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVE_CYCLES)
LD=$(pick SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SMEM)
echo "sq: $SQ | mfma: $MF | lds: $LD" > "$OUT/passes.txt"
K=$TASK; [ "$ENVS" != "4096" ] && K=${TASK}_n$ENVS
run_pass() {   # name, counters..., then the program comes from PROG
  name=$1; shift
  [ -z "$*" ] && return 0
  rocprofv3 --kernel-trace --output-format csv --pmc $@ -d "$OUT/$K/$name" -o p -- $PROG > "$OUT/$K.$name.out" 2> "$OUT/$K.$name.err" || echo "$name pass failed for $K" | tee -a "$OUT/passes.txt"
}
PROG="python3 tools/bench_rollout.py $TASK $ENVS"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$K/stats" -o p -- $PROG > "$OUT/$K.bench.json" 2> "$OUT/$K.stats.err" || echo "stats pass failed for $K" | tee -a "$OUT/passes.txt"
PROG="python3 tools/bench_rollout.py $TASK $ENVS pmc"
run_pass fetch FETCH_SIZE; run_pass write WRITE_SIZE; run_pass sq $SQ; run_pass mfma $MF; run_pass lds $LD
if [ "$TASK" = "combined" ] || [ -n "$RENDER" ]; then
  K=render_n$ENVS
  PROG="python3 tools/bench_render.py $ENVS 32 render_only"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$K/stats" -o p -- $PROG > "$OUT/$K.bench.json" 2> "$OUT/$K.stats.err" || echo "stats pass failed for $K" | tee -a "$OUT/passes.txt"
  run_pass fetch FETCH_SIZE; run_pass write WRITE_SIZE; run_pass sq $SQ; run_pass lds $LD
fi
python3 tools/summarize_learner_pmc.py "$OUT" $ROUND
mkdir -p gpurun_out/${ROUND}_profiles && cp profiles/${ROUND}_learner_* gpurun_out/${ROUND}_profiles/ 2>/dev/null
find "$OUT" -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' ! -name '*.json' ! -name '*.err' ! -name '*.txt' ! -name '*.out' -delete 2>/dev/null || true
