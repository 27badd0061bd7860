set -e
cd $GRAFT_REPO_ROOT
for w in 192 256 384 768; do for st in 4 1; do
  FASTS_HIP_LU_STREAMS=$st FASTS_HIP_LU_FLOW_WAVES=$w timeout -k 10 250 python3 bench.py --no-cpu --implicit 3 --cells 256 --channel 2 1 2 --steps 5 --warmup 2 > gpurun_out/flow/chan_${w}_$st.json
  python3 -c "import json;d=json.load(open('gpurun_out/flow/chan_${w}_$st.json'));print('waves=$w streams=$st ms_per_step %.3f' % d['ms_per_step'])"
done; done
