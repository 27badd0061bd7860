#!/bin/bash
# SQ/GRBM counter pass of the benchmark command: bash profiles/run_sq.sh <tag> [bench args]
set -e
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 180 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/${TAG}_sq -o ${TAG} -- python3 $REPO/bench.py --no-cpu --steps 3 --warmup 1 "$@" > $OUT/${TAG}_sq.log 2>&1
python3 $REPO/profiles/summarise.py pmc $OUT/${TAG}_sq $OUT/${TAG}_sq.csv
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_st -o ${TAG} -- python3 $REPO/bench.py --no-cpu --steps 3 --warmup 1 "$@" > $OUT/${TAG}_st.log 2>&1
python3 $REPO/profiles/summarise.py stats $OUT/${TAG}_st $OUT/${TAG}_st.csv
