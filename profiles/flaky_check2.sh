# repeats the two-ranks-on-one-GPU implicit case under several settings of the one-launch sweeps; counts failures
cd $GRAFT_REPO_ROOT
try() {
  name=$1; shift
  bad=0
  for i in $(seq 1 ${N:-12}); do
    env "$@" timeout -k 10 120 python -u -m pytest tests/test_gpu_mpi.py -x -q --timeout 100 --timeout-method thread -k walls_implicit > gpurun_out/flaky2_${name}_$i.log 2>&1 || bad=$((bad+1))
  done
  echo "$name: failures $bad of ${N:-12}"
}
try waves1 FASTS_HIP_LU_FLOW_WAVES=1
try waves2 FASTS_HIP_LU_FLOW_WAVES=2
try default X=1
