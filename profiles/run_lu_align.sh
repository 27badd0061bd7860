#!/bin/bash
# Units of the block-wise LU sweeps on 128-byte lines (zones whose rows start on a line): parity, then 252^3 with the
# shift on and off, and the 256^3 block (rows not aligned: unchanged).
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/lu_align; mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_lu_blocks.py tests/test_gpu_implicit.py -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for A in 0 1; do
  FASTS_HIP_LU_ALIGN=$A timeout -k 10 200 python3 bench.py --no-cpu --cells 252 --implicit 3 --steps 10 --warmup 2 --lu-threads 64 > $O/c252_A$A.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/c252_A$A.json'));print('252^3 T=64 align=$A ms_per_step', d['ms_per_step'])"
done
timeout -k 10 200 python3 bench.py --no-cpu --cells 252 --implicit 3 --steps 10 --warmup 2 > $O/c252_T1.json 2> $O/err.log
python3 -c "import json;d=json.load(open('$O/c252_T1.json'));print('252^3 T=1 ms_per_step', d['ms_per_step'])"
timeout -k 10 200 python3 bench.py --no-cpu --implicit 3 --steps 10 --warmup 2 --lu-threads 64 > $O/c256.json 2> $O/err.log
python3 -c "import json;d=json.load(open('$O/c256.json'));print('256^3 T=64 ms_per_step', d['ms_per_step'])"
