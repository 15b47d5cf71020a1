#!/bin/bash
# builds and runs the register-fed half-tensor conv ablations on the GPU box: bash tools/micro/conv16w_ablate.sh > gpurun_out/conv16w_ablate.log
#   1 = no weight loads   2 = no halo loads   4 = no epilogue   8 = no MFMAs
set -e
for a in ${ABLS:-0 1 2 4 8 7 15}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iself-supervised-anomaly-detection_amd/csrc -DCONV16W_ABL=$a ${EXTRA} tools/micro/conv16w_ablate.hip -o /tmp/conv16w_abl_$a 2>/dev/null
  timeout -k 5 60 /tmp/conv16w_abl_$a
done
