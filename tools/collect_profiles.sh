#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline figures on the GPU box (run through gpurun):
#   bash tools/collect_profiles.sh [tasks...]
# per task: (1) --kernel-trace --stats of the default (hipGraph) bench run, (2)-(3) HBM traffic (FETCH_SIZE and WRITE_SIZE need
# separate passes), (4) SQ instruction / cycle counters.  Counter passes use --kernel-trace only and launch eagerly (one
# dispatch record per fw_step).  The program after `--` is python3 itself (no shell / env hop under the profiler).
# tools/summarize_profiles.py turns gpurun_out/r02_prof/ into profiles/r02_*.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r02_prof
mkdir -p $OUT
TASKS=${@:-waypoints waypoints_wind objlock combined}
for t in $TASKS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$t/stats -o p -- python3 bench.py --task $t --steps 2000 --warmup 100 --no-cpu-baseline > $OUT/$t.bench.json 2> $OUT/$t.stats.err || echo "stats pass failed for $t"
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/$t/fetch -o p -- python3 bench.py --task $t --steps 256 --warmup 64 --no-cpu-baseline --no-graph > /dev/null 2> $OUT/$t.fetch.err || echo "fetch pass failed for $t"
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/$t/write -o p -- python3 bench.py --task $t --steps 256 --warmup 64 --no-cpu-baseline --no-graph > /dev/null 2> $OUT/$t.write.err || echo "write pass failed for $t"
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $OUT/$t/sq -o p -- python3 bench.py --task $t --steps 256 --warmup 64 --no-cpu-baseline --no-graph > /dev/null 2> $OUT/$t.sq.err || echo "sq pass failed for $t"
  echo "collected $t"
done
python3 tools/summarize_profiles.py $OUT
