#!/bin/bash
# Record of the LU-SGS solve in thread blocks: ms per time step of the two implicit workloads against the number
# of threads, the per-launch trace and the counter passes at 64 threads.  bash profiles/run_profiles_r02_lu.sh
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_lu; mkdir -p $O
: > $O/lu_blocks.txt
for T in 1 16 32 64 128 200; do
  timeout -k 10 200 python3 bench.py --no-cpu --implicit 3 --steps 10 --warmup 2 --lu-threads $T > $O/implicit_T$T.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/implicit_T$T.json'));print('bench.py --implicit 3 --lu-threads $T : %.2f ms per step, stage kernel %.3f ms' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))" >> $O/lu_blocks.txt
done
for T in 1 8 24 64 128 200; do
  timeout -k 10 200 python3 bench.py --no-cpu --naca 384 --scheme roe --steps 20 --warmup 3 --lu-threads $T > $O/naca_T$T.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/naca_T$T.json'));print('bench.py --naca 384 --scheme roe --lu-threads $T : %.2f ms per step' % d['ms_per_step'])" >> $O/lu_blocks.txt
done
for T in 1 64; do
  timeout -k 10 200 python3 bench.py --no-cpu --implicit 3 --perturb 0.2 --steps 6 --warmup 2 --lu-threads $T > $O/curv_T$T.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/curv_T$T.json'));print('bench.py --implicit 3 --perturb 0.2 --lu-threads $T : %.2f ms per step' % d['ms_per_step'])" >> $O/lu_blocks.txt
done
cat $O/lu_blocks.txt
bash profiles/run_lu_trace.sh 64 > $O/trace_T64.txt 2>&1
cd $GRAFT_REPO_ROOT
bash profiles/run_lu_pmc.sh 64 > $O/pmc_T64.txt 2>&1
cd $GRAFT_REPO_ROOT
cp gpurun_out/lu_pmc_T64_256/pmc_*.csv $O/
cp gpurun_out/lu_trace_T64/timeline.txt $O/timeline_T64.txt
