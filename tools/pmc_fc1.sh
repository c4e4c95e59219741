#!/bin/bash
# One rocprofv3 counter pass over tools/fc1_prof.py: tools/pmc_fc1.sh <tag> <counters...>
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcf_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export FC_TORCH=0
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/fc1_prof.py > $OUT/run.log 2>&1 || { echo "pass failed"; tail -3 $OUT/run.log; exit 1; }
grep "fc1" $OUT/run.log
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT fc1_fwd
