#!/bin/bash
# Collects PMC counters for the RoI kernel, one rocprofv3 pass per counter group.
# usage (on the GPU box): bash tools/pmc_roi.sh <tag>
set -u
TAG=${1:-roi}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ROI_REPS=2
i=0
while read -r CTRS; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/roi_prof.py > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; exit 1; }
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE GRBM_TA_BUSY
SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS TA_TA_BUSY_sum TA_BUSY_avr
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
FETCH_SIZE
WRITE_SIZE
LIST
echo "pmc done"
