#!/bin/bash
# Counter passes (one run each) of the implicit bench with the LU-SGS solve in thread blocks.
#   bash profiles/run_lu_pmc.sh <threads>
set -e
T=${1:-64}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lu_pmc_T${T}_${CELLS:-256}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --no-cpu --cells ${CELLS:-256} --implicit 3 --steps 1 --warmup 1 --lu-threads $T"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -o f -- $CMD > $O/pmc_fetch.log 2>&1
python3 $R/profiles/summarise.py pmc $O/prof_fetch $O/pmc_fetch_size.csv
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -o w -- $CMD > $O/pmc_write.log 2>&1
python3 $R/profiles/summarise.py pmc $O/prof_write $O/pmc_write_size.csv
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/prof_sq -o s -- $CMD > $O/pmc_sq.log 2>&1
python3 $R/profiles/summarise.py pmc $O/prof_sq $O/pmc_sq.csv
rm -rf $O/prof_fetch $O/prof_write $O/prof_sq
grep -h "k_lu_" $O/pmc_fetch_size.csv $O/pmc_write_size.csv $O/pmc_sq.csv
