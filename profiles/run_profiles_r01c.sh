#!/bin/bash
# Measurements of the default bench (with the CPU baseline leg), the same command under rocprofv3
# --kernel-trace --stats, and the curvilinear bench.  Run on the GPU box from the repository root.
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 400 python3 bench.py > gpurun_out/r01_final_bench.json 2> gpurun_out/r01_final_bench.err
timeout -k 10 200 python3 bench.py --perturb 0.15 --no-cpu > gpurun_out/r01_final_bench_curvilinear.json 2> gpurun_out/r01_final_curv.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o final -- python3 $R/bench.py --no-cpu > $R/gpurun_out/r01_final_bench_under_rocprof.json 2> $R/gpurun_out/prof_final.log
