#!/bin/bash
# Measurements of the implicit, sub-zone, o3sc and curvilinear paths, each as a bench line and the same command under
# rocprofv3 --kernel-trace --stats (summaries only are kept).  Run on the GPU box from the repository root:
#   bash profiles/run_profiles_r03b.sh <tag>
set -e
TAG=${1:-r03b}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
run() {   # name, bench arguments
  local name=$1; shift
  cd $R
  timeout -k 10 300 python3 bench.py --no-cpu "$@" > $OUT/${name}_bench.json 2> $OUT/${name}_bench.err
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o st -- python3 $R/bench.py --no-cpu "$@" > $OUT/${name}_under_rocprof.json 2> $OUT/${name}_prof.log
  python3 $R/profiles/summarise.py stats $OUT/prof_$name $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/prof_$name
}
run implicit3 --implicit 3 --steps 10 --warmup 3
run implicit6_local32 --implicit 6 --local-newton 32 --epsi-newton 0.3 --steps 14 --warmup 16
run o3sc --slope o3sc --steps 10 --warmup 3
run o3sc_curv --slope o3sc --perturb 0.2 --steps 5 --warmup 2
run curv --perturb 0.2 --steps 20 --warmup 3
ls $OUT
