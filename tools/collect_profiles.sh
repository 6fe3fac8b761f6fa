#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline figures on the GPU box (run through gpurun):
#   bash tools/collect_profiles.sh [tasks...]
# per task: (1) --kernel-trace --stats of the default (hipGraph) bench run, (2)-(3) HBM traffic (FETCH_SIZE and WRITE_SIZE need
# separate passes), (4) SQ instruction / cycle counters.  Counter passes use --kernel-trace only and launch eagerly (one
# dispatch record per fw_step).  The program after `--` is python3 itself (no shell / env hop under the profiler).
# tools/summarize_profiles.py turns gpurun_out/${ROUND}_prof/ into profiles/${ROUND}_*.  ROUND defaults to r03.
# ENVS=<n> LANES=<1|4|8> collect the same four passes at another env count / lane mapping into .../<task>_n<ENVS>/ (the
# large-N roofline evidence: FETCH/WRITE_SIZE and SQ_INSTS_VALU at N = 2^20 on the one-lane-per-env mapping).
set -o pipefail
export TMPDIR=/tmp
ROUND=${ROUND:-r04}
OUT=gpurun_out/${ROUND}_prof
ENVS=${ENVS:-4096}
EXTRA="--envs-per-gpu $ENVS"
SFX=""
STEPS=2000; CSTEPS=256; CWARM=64
if [ "$ENVS" != "4096" ]; then SFX="_n$ENVS"; fi
if [ "$ENVS" -gt 65536 ]; then STEPS=200; CSTEPS=40; CWARM=8; fi
if [ -n "$LANES" ]; then export FWSIM_LANES_PER_ENV=$LANES; fi
mkdir -p $OUT
TASKS=${@:-waypoints waypoints_wind objlock combined}
for t in $TASKS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$t$SFX/stats -o p -- python3 bench.py --task $t --steps $STEPS --warmup 100 --repeats 3 --no-cpu-baseline $EXTRA > $OUT/$t$SFX.bench.json 2> $OUT/$t$SFX.stats.err || echo "stats pass failed for $t"
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/$t$SFX/fetch -o p -- python3 bench.py --task $t --steps $CSTEPS --warmup $CWARM --repeats 1 --no-cpu-baseline --no-graph $EXTRA > /dev/null 2> $OUT/$t$SFX.fetch.err || echo "fetch pass failed for $t"
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/$t$SFX/write -o p -- python3 bench.py --task $t --steps $CSTEPS --warmup $CWARM --repeats 1 --no-cpu-baseline --no-graph $EXTRA > /dev/null 2> $OUT/$t$SFX.write.err || echo "write pass failed for $t"
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $OUT/$t$SFX/sq -o p -- python3 bench.py --task $t --steps $CSTEPS --warmup $CWARM --repeats 1 --no-cpu-baseline --no-graph $EXTRA > /dev/null 2> $OUT/$t$SFX.sq.err || echo "sq pass failed for $t"
  echo "collected $t"
done
python3 tools/summarize_profiles.py $OUT $ROUND
# the GPU box returns gpurun_out/ only: leave a copy of the summaries there (the raw counter dumps are trimmed to the csv files)
mkdir -p gpurun_out/${ROUND}_profiles && cp profiles/${ROUND}_* gpurun_out/${ROUND}_profiles/ 2>/dev/null
find $OUT -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' ! -name '*.json' ! -name '*.err' -delete 2>/dev/null || true
