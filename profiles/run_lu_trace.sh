#!/bin/bash
# Kernel trace of the implicit bench with the LU-SGS solve in thread blocks: per-launch durations of the sweeps against
# the number of workgroups.   bash profiles/run_lu_trace.sh <threads>
set -e
T=${1:-64}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lu_trace_T$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format rocpd -d $O/prof -o tr -- python3 $R/bench.py --no-cpu --implicit 3 --steps 2 --warmup 1 --lu-threads $T > $O/bench.json 2> $O/prof.log
DB=$(find $O/prof -name '*.db' | head -1)
python3 $R/profiles/lu_timeline.py $DB > $O/timeline.txt
rm -rf $O/prof
cat $O/timeline.txt
