#!/bin/bash
# Run on the GPU box: bench line, rocprofv3 kernel stats of the same command, and the
# HBM-traffic PMC passes for the dominant kernel.  Results land in gpurun_out/<tag>/.
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/stats.log 2>&1 || { echo stats failed; exit 1; }
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE TA_BUSY_avr"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout -s KILL 180 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_$N.log 2>&1 || { echo pmc $N failed; exit 1; }
done
echo collected
