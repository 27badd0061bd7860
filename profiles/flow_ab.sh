# A/B of the LU-SGS sweeps: one launch per hyperplane (flow=0) against one launch per sweep with N waves
# usage: bash profiles/flow_ab.sh "<flow> <waves>" ...   (bench.py --implicit 3, 256^3)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/flow
for cfg in "$@"; do
  set -- $cfg
  export FASTS_HIP_LU_FLOW=$1 FASTS_HIP_LU_FLOW_WAVES=$2
  timeout -k 10 200 python3 bench.py --implicit 3 --no-cpu --steps 8 --warmup 2 > gpurun_out/flow/imp3_$1_$2.json
  python3 -c "import json;d=json.load(open('gpurun_out/flow/imp3_$1_$2.json'));print('flow=$1 waves=$2 ms_per_step', d['ms_per_step'])"
done
