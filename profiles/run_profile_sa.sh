#!/bin/bash
# Spalart-Allmaras stage kernel on the Cartesian box (bench.py --sa): the bench line and the same command under
# rocprofv3 --kernel-trace --stats.   bash profiles/run_profile_sa.sh <tag>
set -e
TAG=${1:-r03d_sa}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 300 python3 bench.py --sa --no-cpu --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o st -- python3 $R/bench.py --sa --no-cpu --steps 10 --warmup 2 > $OUT/bench_under_rocprof.json 2> $OUT/prof_stats.log
python3 $R/profiles/summarise.py stats $OUT/prof_stats $OUT/kernel_stats.csv
rm -rf $OUT/prof_stats
cat $OUT/bench.json
