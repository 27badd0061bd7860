#!/bin/bash
# Curvilinear / Spalart-Allmaras stage kernels after the change of their addressing (scalar base + 32-bit lane offset):
# parity tests of the zones that run k_stage_full, then ms per step of the bench cases that use it.
set -e
mkdir -p gpurun_out/addr
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_sa.py tests/test_gpu_naca.py tests/test_gpu_homo.py tests/test_gpu_o3sc.py tests/test_gpu_combos.py tests/test_gpu_local_newton.py tests/test_gpu_implicit.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/addr/tests.log 2>&1
tail -2 gpurun_out/addr/tests.log
run() { name=$1; shift; timeout -k 10 200 python bench.py --no-cpu --steps 20 --warmup 3 "$@" > gpurun_out/addr/$name.json 2> gpurun_out/addr/$name.err; python - <<PY
import json
d=json.load(open("gpurun_out/addr/$name.json"))
print("$name", round(d["ms_per_step"],3), "ms/step; stage", round(d["roofline"]["avg_launch_ms"],4), "ms; frac", round(d["roofline"]["frac"],3))
PY
}
run curv --perturb 0.2
run curv_ausm --perturb 0.2 --scheme ausmpred
run curv_roe --perturb 0.2 --scheme roe
run sa_curv --sa --perturb 0.2
run naca --naca 384 --scheme roe
run curv_impl --perturb 0.2 --implicit 3
run sa --sa
run curv_euler_roe --perturb 0.2 --scheme roe
