#!/bin/bash
# Measurements of the default bench (TGV 256^3 senseur RK3): the bench line, the same command under
# rocprofv3 --kernel-trace --stats, and separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ set).  Run on the GPU box
# from the repository root:   bash profiles/run_profiles_r03.sh <tag>
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o st -- python3 $R/bench.py --no-cpu --steps 20 --warmup 3 > $OUT/bench_under_rocprof.json 2> $OUT/prof_stats.log
python3 $R/profiles/summarise.py stats $OUT/prof_stats $OUT/kernel_stats.csv
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o f -- python3 $R/bench.py --no-cpu --steps 3 --warmup 1 > $OUT/pmc_fetch.log 2>&1
python3 $R/profiles/summarise.py pmc $OUT/prof_fetch $OUT/pmc_fetch_size.csv
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o w -- python3 $R/bench.py --no-cpu --steps 3 --warmup 1 > $OUT/pmc_write.log 2>&1
python3 $R/profiles/summarise.py pmc $OUT/prof_write $OUT/pmc_write_size.csv
python3 $R/profiles/summarise.py traffic $OUT/pmc_fetch_size.csv $OUT/pmc_write_size.csv $OUT/traffic_senseur_256.json
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/prof_sq -o s -- python3 $R/bench.py --no-cpu --steps 3 --warmup 1 > $OUT/pmc_sq.log 2>&1
python3 $R/profiles/summarise.py pmc $OUT/prof_sq $OUT/pmc_sq.csv
rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write $OUT/prof_sq
cat $OUT/bench.json
