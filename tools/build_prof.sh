#!/bin/bash
# Dev builds of the library with the per-wave cycle accounting compiled in (tools/wave_profile.py):
#   bash tools/build_prof.sh          -> tools/_build/libfwsim_prof.so      (-DFW_PROFILE)
#   bash tools/build_prof.sh phases   -> tools/_build/libfwsim_prof_ph.so   (-DFW_PROFILE -DFW_PROFILE_PHASES)
#   bash tools/build_prof.sh ppo      -> tools/_build/libfwsim_ppoprof.so   (-DFW_PPO_PROF)
# Like the product build (pyflyt_drone_amd/_lib.py) each one is checked with tools/check_isa.py: the extra live registers of
# the accounting shift the register allocation, and a build with a spill store ahead of an exec restore computes garbage in
# the lanes that skipped the branch (the phases build of 2026-10-04 faulted on exactly that).  Such a build is deleted unless
# KEEP_SUSPECT=1 (the check is conservative: it cannot tell whether the lanes outside the branch still need the value).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/tools/_build; mkdir -p $OUT
if [ "$1" = "phases" ]; then NAME=libfwsim_prof_ph.so; DEFS="-DFW_PROFILE -DFW_PROFILE_PHASES";
elif [ "$1" = "ppo" ]; then NAME=libfwsim_ppoprof.so; DEFS="-DFW_PPO_PROF";      # cycle stamps inside fw_ppo_update (tools/prof_ppo.py)
else NAME=libfwsim_prof.so; DEFS="-DFW_PROFILE"; fi
TMP=$(mktemp -d); trap "rm -rf $TMP" EXIT
(cd $TMP && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -save-temps $DEFS -o $OUT/$NAME $ROOT/pyflyt-drone_amd/csrc/fwsim.hip)
if ! python3 $ROOT/tools/check_isa.py $TMP/*gfx950*.s > $TMP/check.txt; then
  cat $TMP/check.txt
  if [ -z "$KEEP_SUSPECT" ]; then rm -f $OUT/$NAME; echo "$NAME NOT built (KEEP_SUSPECT=1 keeps it: do not run the kernels named above)"; exit 1; fi
  echo "WARNING: $NAME kept although the kernels named above are suspect"
fi
echo "built $OUT/$NAME"
