#!/bin/bash
# Block-wise LU sweeps: every block's workgroups on one XCD (default) against the plain order (FASTS_HIP_LU_ALIGN bit 2).
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/lu_xcd; mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_lu_blocks.py -x -q --timeout 200 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for A in 5 1; do
for T in 64 128; do
  FASTS_HIP_LU_ALIGN=$A timeout -k 10 200 python3 bench.py --no-cpu --implicit 3 --steps 10 --warmup 2 --lu-threads $T > $O/c256_A${A}_T$T.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/c256_A${A}_T$T.json'));print('256^3 switch=$A T=$T ms_per_step', d['ms_per_step'])"
done
  FASTS_HIP_LU_ALIGN=$A timeout -k 10 200 python3 bench.py --no-cpu --cells 252 --implicit 3 --steps 10 --warmup 2 --lu-threads 64 > $O/c252_A${A}.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/c252_A${A}.json'));print('252^3 switch=$A T=64 ms_per_step', d['ms_per_step'])"
  FASTS_HIP_LU_ALIGN=$A timeout -k 10 200 python3 bench.py --no-cpu --naca 384 --scheme roe --steps 20 --warmup 3 --lu-threads 24 > $O/naca_A${A}.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/naca_A${A}.json'));print('naca switch=$A T=24 ms_per_step', d['ms_per_step'])"
  FASTS_HIP_LU_ALIGN=$A timeout -k 10 200 python3 bench.py --no-cpu --implicit 3 --nb-relax 2 --steps 6 --warmup 2 --lu-threads 64 > $O/ssor_A${A}.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/ssor_A${A}.json'));print('256^3 nb_relax=2 switch=$A T=64 ms_per_step', d['ms_per_step'])"
done
