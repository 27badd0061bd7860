#!/bin/bash
# Block-wise LU solve: parity tests, then timings of the implicit workloads for several thread counts and unit lengths.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/lu_blocks
mkdir -p $O
for U in ${UNITS:-8 16}; do
  export FASTS_HIP_LU_UNIT=$U
  timeout -k 10 300 python -u -m pytest tests/test_gpu_lu_blocks.py tests/test_gpu_implicit.py -x -q --timeout 120 --timeout-method thread > $O/tests_U$U.log 2>&1 || { tail -30 $O/tests_U$U.log; exit 1; }
  tail -1 $O/tests_U$U.log
  for T in ${THREADS:-1 16 64 128}; do
    timeout -k 10 200 python3 bench.py --no-cpu --implicit 3 --steps 10 --warmup 2 --lu-threads $T > $O/implicit_U${U}_T$T.json 2> $O/implicit_U${U}_T$T.err
    python3 -c "import json;d=json.load(open('$O/implicit_U${U}_T$T.json'));print('U=$U T=$T ms_per_step', d['ms_per_step'])"
  done
  for T in ${NACA_THREADS:-1 8 24}; do
    timeout -k 10 200 python3 bench.py --no-cpu --naca 384 --scheme roe --steps 20 --warmup 3 --lu-threads $T > $O/naca_U${U}_T$T.json 2> $O/naca_U${U}_T$T.err
    python3 -c "import json;d=json.load(open('$O/naca_U${U}_T$T.json'));print('naca U=$U T=$T ms_per_step', d['ms_per_step'])"
  done
done
