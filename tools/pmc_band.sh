#!/bin/bash
# Two SQ counter passes over the band-resident RoI forward kernel: bash tools/pmc_band.sh <tag>
set -u
TAG=${1:-band}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ROI_REPS=2 FRCNN_ROI_BAND_MIN_WG=1 ROI_BAND=1   # read once, when the library is loaded
i=0
while read -r CTRS; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/roi_prof.py > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; exit 1; }
done <<'LIST'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
LIST
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT roi_fwd_band
