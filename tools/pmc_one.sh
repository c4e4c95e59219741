#!/bin/bash
# One rocprofv3 counter pass over tools/roi_prof.py: tools/pmc_one.sh <tag> <counters...>
# (set FRCNN_HIP_LIB to profile an experiment build)
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc1_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ROI_REPS=2
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/roi_prof.py > $OUT/run.log 2>&1 || { echo "pass failed"; tail -3 $OUT/run.log; exit 1; }
grep "roi fwd" $OUT/run.log
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT roi_fwd
