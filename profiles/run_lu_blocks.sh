#!/bin/bash
# Block-wise LU solve: parity tests of the implicit scheme, then timings for a few thread counts.
#   THREADS="1 64" NACA_THREADS="1 24" RELAX=2 bash profiles/run_lu_blocks.sh
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/lu_blocks
mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_lu_blocks.py tests/test_gpu_implicit.py tests/test_gpu_naca.py tests/test_gpu_combos.py tests/test_gpu_sa.py -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
R=${RELAX:-1}
for T in ${THREADS:-1 64}; do
  timeout -k 10 200 python3 bench.py --no-cpu --implicit 3 --steps 10 --warmup 2 --lu-threads $T --nb-relax $R > $O/implicit_R${R}_T$T.json 2> $O/implicit_R${R}_T$T.err
  python3 -c "import json;d=json.load(open('$O/implicit_R${R}_T$T.json'));print('nb_relax=$R T=$T ms_per_step', d['ms_per_step'])"
done
for T in ${NACA_THREADS:-1 24}; do
  timeout -k 10 200 python3 bench.py --no-cpu --naca 384 --scheme roe --steps 20 --warmup 3 --lu-threads $T --nb-relax $R > $O/naca_R${R}_T$T.json 2> $O/naca_R${R}_T$T.err
  python3 -c "import json;d=json.load(open('$O/naca_R${R}_T$T.json'));print('naca nb_relax=$R T=$T ms_per_step', d['ms_per_step'])"
done
