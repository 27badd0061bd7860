#!/bin/bash
# Profiles of the benchmark command on the GPU box (run through gpurun from the repo root):
#   bash profiles/run_profiles.sh <tag>
# One kernel-trace/stats run, then the counter passes, each in a run of its own.  Raw output
# goes to gpurun_out/<tag>_*/, the per-kernel summaries to gpurun_out/<tag>_*.csv; copy the
# summaries into profiles/ to keep them.
set -e
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $REPO/bench.py --no-cpu"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o ${TAG} -- $BENCH --steps 10 --warmup 2 > $OUT/${TAG}_stats.log 2>&1
python3 $REPO/profiles/summarise.py stats $OUT/${TAG}_stats $OUT/${TAG}_kernel_stats.csv
grep '"metric"' $OUT/${TAG}_stats.log > $OUT/${TAG}_bench_under_rocprof.json || true
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  NAME=$(echo $PASS | cut -d' ' -f1)
  timeout -s KILL 180 rocprofv3 --pmc $PASS --output-format csv -d $OUT/${TAG}_pmc_$NAME -o ${TAG} -- $BENCH --steps 3 --warmup 1 > $OUT/${TAG}_pmc_$NAME.log 2>&1
  python3 $REPO/profiles/summarise.py pmc $OUT/${TAG}_pmc_$NAME $OUT/${TAG}_pmc_$NAME.csv
done
python3 $REPO/profiles/summarise.py traffic $OUT/${TAG}_pmc_FETCH_SIZE.csv $OUT/${TAG}_pmc_WRITE_SIZE.csv $OUT/${TAG}_traffic.json
echo done
