#!/bin/bash
# A/B of stage-kernel builds on the default bench: bash profiles/ab_stage.sh <tag> <lib> [<lib> ...]
# (libraries relative to fast_amd/csrc); writes gpurun_out/<tag>/<lib>.json
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for L in "$@"; do
  FASTS_HIP_LIB=$R/fast_amd/csrc/$L timeout -k 10 200 python3 bench.py --no-cpu --steps 40 --warmup 5 > $OUT/$L.json 2> $OUT/$L.err || { echo "FAILED $L"; tail -5 $OUT/$L.err; exit 1; }
  python3 - "$OUT/$L.json" "$L" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-28s ms/step %.3f  stage %.4f ms  frac %.3f  value %.0f" % (sys.argv[2], d["ms_per_step"], r["avg_launch_ms"], r["frac"], d["value"]))
PY
done
