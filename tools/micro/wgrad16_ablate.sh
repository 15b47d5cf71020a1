#!/bin/bash
# builds and runs the weight-gradient ablations on the GPU box: bash tools/micro/wgrad16_ablate.sh > gpurun_out/wgrad16_ablate.log
#   1 = no global loads   2 = no MFMAs   4 = no LDS tile writes
set -e
for a in ${ABLS:-0 1 2 5 7}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iself-supervised-anomaly-detection_amd/csrc -DWG16_ABL=$a ${EXTRA} tools/micro/wgrad16_ablate.hip -o /tmp/wg16_abl_$a 2>/dev/null
  timeout -k 5 60 /tmp/wg16_abl_$a
done
