set -e
cd $GRAFT_REPO_ROOT
FASTS_HIP_LU_FLOW_UNIT=4 timeout -k 10 300 python -u -m pytest tests/test_gpu_implicit.py -x -q --timeout 200 --timeout-method thread 2>&1 | tail -3
for u in 8 4; do for w in 512 768 1024; do
  FASTS_HIP_LU_FLOW_UNIT=$u FASTS_HIP_LU_FLOW_WAVES=$w timeout -k 10 200 python3 bench.py --implicit 3 --no-cpu --steps 8 --warmup 2 > gpurun_out/flow/unit_${u}_$w.json
  python3 -c "import json;d=json.load(open('gpurun_out/flow/unit_${u}_$w.json'));print('unit=$u waves=$w ms_per_step', d['ms_per_step'])"
done; done
