#!/bin/bash
# The fw_render part of tools/collect_learner_pmc.sh alone (same passes, same output layout, so tools/summarize_learner_pmc.py updates
# only the `render_n<envs>` entry of profiles/${ROUND}_learner_pmc.json), plus launch statistics at 64 x 64 and 128 x 128 pixels:
#    bash tools/collect_render_pmc.sh [envs]
set -o pipefail
export TMPDIR=/tmp
ROUND=${ROUND:-r05}
ENVS=${1:-4096}
OUT=$PWD/gpurun_out/${ROUND}_learner; mkdir -p "$OUT"
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1 || true
pick() { r=""; for c in "$@"; do grep -qw "$c" "$OUT/counters_list.txt" && r="$r $c"; done; echo $r; }
SQ=$(pick SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES)
LD=$(pick SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SMEM)
K=render_n$ENVS
run_pass() {
  name=$1; shift
  [ -z "$*" ] && return 0
  rocprofv3 --kernel-trace --output-format csv --pmc $@ -d "$OUT/$K/$name" -o p -- $PROG > "$OUT/$K.$name.out" 2> "$OUT/$K.$name.err" || echo "$name pass failed for $K" | tee -a "$OUT/passes.txt"
}
PROG="python3 tools/bench_render.py $ENVS 32 render_only"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$K/stats" -o p -- $PROG > "$OUT/$K.bench.json" 2> "$OUT/$K.stats.err" || echo "stats pass failed for $K" | tee -a "$OUT/passes.txt"
run_pass fetch FETCH_SIZE; run_pass write WRITE_SIZE; run_pass sq $SQ; run_pass lds $LD
mkdir -p profiles
: > profiles/${ROUND}_render_bench.jsonl
for res in 32 64 128; do
  python3 tools/bench_render.py $ENVS $res render_only >> profiles/${ROUND}_render_bench.jsonl 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/render_res$res" -o p -- python3 tools/bench_render.py $ENVS $res render_only > /dev/null 2> "$OUT/render_res$res.err" \
    && cp "$OUT/render_res$res/p_kernel_stats.csv" profiles/${ROUND}_render_${res}px_kernel_stats.csv
done
rm -rf "$OUT"/render_res*
python3 tools/summarize_learner_pmc.py "$OUT" $ROUND > "$OUT/summary.txt"
mkdir -p gpurun_out/${ROUND}_profiles && cp profiles/${ROUND}_learner_pmc.json profiles/${ROUND}_render_* profiles/${ROUND}_rollout_render_* gpurun_out/${ROUND}_profiles/ 2>/dev/null
find "$OUT" -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' ! -name '*.json' ! -name '*.err' ! -name '*.txt' ! -name '*.out' -delete 2>/dev/null || true
grep -A4 "render" "$OUT/summary.txt"; cat profiles/${ROUND}_render_bench.jsonl | cut -c1-110
