# one launch per sweep for the boxes of implicit_local and (FASTS_HIP_LU_FLOW_BLOCKS=1) for the thread blocks
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/flow
timeout -k 10 300 python -u -m pytest tests/test_gpu_local_newton.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -3
FASTS_HIP_LU_FLOW_BLOCKS=1 timeout -k 10 300 python -u -m pytest tests/test_gpu_lu_blocks.py tests/test_gpu_naca.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -3
b() { name=$1; shift; timeout -k 10 250 python3 bench.py --no-cpu "$@" > gpurun_out/flow/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/flow/$name.json'));print('$name ms_per_step %.3f' % d['ms_per_step'])"; }
FASTS_HIP_LU_FLOW=0 b local32_e03_hyper --implicit 6 --local-newton 32 --epsi-newton 0.3 --steps 14 --warmup 16
b local32_e03_flow --implicit 6 --local-newton 32 --epsi-newton 0.3 --steps 14 --warmup 16
FASTS_HIP_LU_FLOW=0 b local32_e01_hyper --implicit 6 --local-newton 32 --epsi-newton 0.1 --steps 14 --warmup 16
b local32_e01_flow --implicit 6 --local-newton 32 --epsi-newton 0.1 --steps 14 --warmup 16
b implicit6_flow --implicit 6 --steps 8 --warmup 2
for T in 16 64 128; do
  b blocks${T}_hyper --implicit 3 --lu-threads $T --steps 8 --warmup 2
  FASTS_HIP_LU_FLOW_BLOCKS=1 b blocks${T}_flow --implicit 3 --lu-threads $T --steps 8 --warmup 2
done
b naca24_hyper --naca 384 --scheme roe --lu-threads 24 --steps 20 --warmup 3
FASTS_HIP_LU_FLOW_BLOCKS=1 b naca24_flow --naca 384 --scheme roe --lu-threads 24 --steps 20 --warmup 3
