#!/bin/bash
# Launch-latency experiment: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG) for the latency-bound LU sweeps.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/kernarg; mkdir -p $O
for V in 0 1; do
  HIP_FORCE_DEV_KERNARG=$V timeout -k 10 200 python3 bench.py --no-cpu --implicit 3 --steps 10 --warmup 2 > $O/implicit_$V.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/implicit_$V.json'));print('HIP_FORCE_DEV_KERNARG=$V implicit ms_per_step', d['ms_per_step'])"
  HIP_FORCE_DEV_KERNARG=$V timeout -k 10 200 python3 bench.py --no-cpu --naca 384 --scheme roe --steps 20 --warmup 3 > $O/naca_$V.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/naca_$V.json'));print('HIP_FORCE_DEV_KERNARG=$V naca ms_per_step', d['ms_per_step'])"
  HIP_FORCE_DEV_KERNARG=$V timeout -k 10 200 python3 bench.py --no-cpu --naca 384 --scheme roe --steps 20 --warmup 3 --lu-threads 200 > $O/naca200_$V.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/naca200_$V.json'));print('HIP_FORCE_DEV_KERNARG=$V naca T=200 ms_per_step', d['ms_per_step'])"
  HIP_FORCE_DEV_KERNARG=$V timeout -k 10 200 python3 bench.py --no-cpu --steps 50 --warmup 5 > $O/explicit_$V.json 2> $O/err.log
  python3 -c "import json;d=json.load(open('$O/explicit_$V.json'));print('HIP_FORCE_DEV_KERNARG=$V explicit ms_per_step', d['ms_per_step'])"
done
timeout -k 10 200 python3 bench.py --no-cpu --implicit 3 --steps 10 --warmup 2 > $O/implicit_unset.json 2> $O/err.log
python3 -c "import json;d=json.load(open('$O/implicit_unset.json'));print('unset implicit ms_per_step', d['ms_per_step'])"
