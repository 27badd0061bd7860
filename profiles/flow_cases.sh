# the LU-SGS sweeps as one launch per hyperplane (FASTS_HIP_LU_FLOW=0) and as one launch per sweep (=1) on the other
# implicit workloads of bench.py; prints ms per time step
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/flow
run() {
  name=$1; shift
  for f in 0 1; do
    FASTS_HIP_LU_FLOW=$f timeout -k 10 250 python3 bench.py --no-cpu "$@" > gpurun_out/flow/${name}_$f.json
    python3 -c "import json;d=json.load(open('gpurun_out/flow/${name}_$f.json'));print('$name flow=$f ms_per_step %.3f' % d['ms_per_step'])"
  done
}
run naca384_roe --naca 384 --scheme roe --steps 20 --warmup 3
run curv_imp3 --implicit 3 --perturb 0.2 --steps 6 --warmup 2
run cells128_imp3 --implicit 3 --cells 128 --steps 10 --warmup 3
run channel_512 --implicit 3 --cells 256 --channel 2 1 2 --steps 5 --warmup 2
