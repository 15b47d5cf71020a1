#!/bin/bash
# builds and runs the half-tensor halo-conv ablations on the GPU box: bash tools/micro/conv16_ablate.sh > gpurun_out/conv16_ablate.log
set -e
for a in ${ABLS:-0 1 2 3 4 7 8 16 24 32 63}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iself-supervised-anomaly-detection_amd/csrc -DCONV16_ABL=$a ${EXTRA} tools/micro/conv16_ablate.hip -o /tmp/conv16_abl_$a 2>/dev/null
  timeout -k 5 60 /tmp/conv16_abl_$a
done
