#!/bin/bash
# Implicit workloads after a change of the sweep kernels: parity, then the usual timings.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/lu_final; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_lu_blocks.py tests/test_gpu_implicit.py tests/test_gpu_naca.py tests/test_gpu_combos.py tests/test_gpu_sa.py tests/test_gpu_homo.py tests/test_gpu_2d.py tests/test_gpu_golden.py tests/test_gpu_mpi.py -x -q --timeout 200 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
: > $O/times.txt
for T in 1 64 128; do
  timeout -k 10 200 python3 bench.py --no-cpu --implicit 3 --steps 10 --warmup 2 --lu-threads $T > $O/implicit_T$T.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/implicit_T$T.json'));print('bench.py --implicit 3 --lu-threads $T : %.2f ms per step' % d['ms_per_step'])" >> $O/times.txt
done
timeout -k 10 200 python3 bench.py --no-cpu --cells 252 --implicit 3 --steps 10 --warmup 2 --lu-threads 64 > $O/c252.json 2> $O/err.log
python3 -c "import json;d=json.load(open('$O/c252.json'));print('bench.py --cells 252 --implicit 3 --lu-threads 64 : %.2f ms per step' % d['ms_per_step'])" >> $O/times.txt
for T in 1 24 200; do
  timeout -k 10 200 python3 bench.py --no-cpu --naca 384 --scheme roe --steps 20 --warmup 3 --lu-threads $T > $O/naca_T$T.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/naca_T$T.json'));print('bench.py --naca 384 --scheme roe --lu-threads $T : %.2f ms per step' % d['ms_per_step'])" >> $O/times.txt
done
for T in 1 64; do
  timeout -k 10 200 python3 bench.py --no-cpu --implicit 3 --nb-relax 2 --steps 6 --warmup 2 --lu-threads $T > $O/ssor_T$T.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/ssor_T$T.json'));print('bench.py --implicit 3 --nb-relax 2 --lu-threads $T : %.2f ms per step' % d['ms_per_step'])" >> $O/times.txt
done
cat $O/times.txt
