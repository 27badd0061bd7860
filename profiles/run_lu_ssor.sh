#!/bin/bash
# LU-SSOR: parity tests of the implicit scheme, then timings with 2 relaxations.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/lu_ssor; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_lu_blocks.py tests/test_gpu_implicit.py tests/test_gpu_naca.py tests/test_gpu_combos.py tests/test_gpu_sa.py tests/test_gpu_homo.py tests/test_gpu_2d.py tests/test_gpu_golden.py -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for T in 1 64; do
  timeout -k 10 200 python3 bench.py --no-cpu --implicit 3 --steps 10 --warmup 2 --lu-threads $T --nb-relax 2 > $O/implicit_R2_T$T.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/implicit_R2_T$T.json'));print('256^3 nb_relax=2 T=$T ms_per_step', d['ms_per_step'])"
done
for T in 1 24; do
  timeout -k 10 200 python3 bench.py --no-cpu --naca 384 --scheme roe --steps 20 --warmup 3 --lu-threads $T --nb-relax 2 > $O/naca_R2_T$T.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/naca_R2_T$T.json'));print('naca nb_relax=2 T=$T ms_per_step', d['ms_per_step'])"
done
