#!/bin/bash
# rocprofv3 kernel statistics of the curvilinear and implicit workloads (same commands as the bench lines
# quoted in the design notes).  Run on the GPU box from the repository root; summaries land in gpurun_out/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in curv impl; do
  if [ $tag = curv ]; then ARGS="--perturb 0.15 --steps 5 --warmup 1 --no-cpu"; else ARGS="--implicit 3 --steps 2 --warmup 1 --no-cpu --scheme ausmpred"; fi
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_$tag.log 2>&1
done
