#!/bin/bash
# Builds experiment variants of the library: tools/build_variants.sh "<name> <hipcc -D flags>" ...
# Each lands in objectdetection_amd/lib/exp_<name>.so; select one with FRCNN_HIP_LIB.
set -e
cd "$(dirname "$0")/../objectdetection_amd/csrc"
for spec in "$@"; do
  set -- $spec
  name=$1; shift
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC "$@" -c frcnn_roi.hip -o /tmp/roi_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC frcnn_abi.o frcnn_boxes.o frcnn_sample.o frcnn_proposal.o frcnn_topk.o frcnn_roi_band.o frcnn_detect.o frcnn_loss.o frcnn_input.o /tmp/roi_$name.o -o ../lib/exp_$name.so
done
