# LU-SSOR (nb_relax 2) and lu_match workloads: one launch per hyperplane (FASTS_HIP_LU_FLOW=0) against one per sweep
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/flow
b() { name=$1; shift; timeout -k 10 250 python3 bench.py --no-cpu "$@" > gpurun_out/flow/$name.json; python3 -c "import json;d=json.load(open('gpurun_out/flow/$name.json'));print('$name ms_per_step %.3f' % d['ms_per_step'])"; }
FASTS_HIP_LU_FLOW=0 b ssor2_hyper --implicit 3 --nb-relax 2 --steps 5 --warmup 2
b ssor2_flow --implicit 3 --nb-relax 2 --steps 5 --warmup 2
FASTS_HIP_LU_FLOW=0 b naca_ssor2_hyper --naca 384 --scheme roe --nb-relax 2 --steps 20 --warmup 3
b naca_ssor2_flow --naca 384 --scheme roe --nb-relax 2 --steps 20 --warmup 3
FASTS_HIP_LU_FLOW=0 b naca_ssor2_T24_hyper --naca 384 --scheme roe --nb-relax 2 --lu-threads 24 --steps 20 --warmup 3
b naca_ssor2_T24_flow --naca 384 --scheme roe --nb-relax 2 --lu-threads 24 --steps 20 --warmup 3
FASTS_HIP_LU_FLOW=0 b ssor2_T64_hyper --implicit 3 --nb-relax 2 --lu-threads 64 --steps 5 --warmup 2
FASTS_HIP_LU_FLOW_BLOCKS=1 b ssor2_T64_flow --implicit 3 --nb-relax 2 --lu-threads 64 --steps 5 --warmup 2
