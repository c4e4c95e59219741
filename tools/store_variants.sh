#!/bin/bash
# Runs the RoI stress test against each experiment build of the store (tools/build_variants.sh
# "stN -DFRCNN_STORE_VARIANT=N"); a failing assertion (exit 1) is a result, anything else stops.
cd $GRAFT_REPO_ROOT
for v in 1 2 3 4 5; do
  FRCNN_HIP_LIB=$PWD/objectdetection_amd/lib/exp_st$v.so timeout -k 10 200 python -u -m pytest tests/test_gpu_roi_stress.py -q -k "per_roi" --timeout 120 --timeout-method thread > gpurun_out/store_v$v.log 2>&1
  rc=$?
  echo "variant $v: exit $rc: $(tail -1 gpurun_out/store_v$v.log)"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping"; exit $rc; fi
done
