#!/bin/bash
# One rocprofv3 counter pass over tools/bwd_prof.py: tools/pmc_bwd.sh <tag> <counters...>
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcb_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ROI_REPS=2
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bwd_prof.py > $OUT/run.log 2>&1 || { echo "pass failed"; tail -3 $OUT/run.log; exit 1; }
grep "roi bwd" $OUT/run.log
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT roi_bwd
