#!/bin/bash
# Counter passes of the one-launch LU-SGS sweeps (bench.py --implicit 3, 256^3), one rocprofv3 --pmc run each:
#   bash profiles/run_flow_pmc.sh <tag>
set -e
TAG=${1:-r03c_flow_pmc}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {   # name, counters
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/prof_$name -o p -- python3 $R/bench.py --no-cpu --implicit 3 --steps 2 --warmup 1 > $OUT/pmc_$name.log 2>&1
  python3 $R/profiles/summarise.py pmc $OUT/prof_$name $OUT/pmc_$name.csv
  rm -rf $OUT/prof_$name
}
pass sq GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
pass fetch FETCH_SIZE
pass write WRITE_SIZE
ls $OUT
