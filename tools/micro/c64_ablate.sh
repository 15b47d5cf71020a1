#!/bin/bash
# builds and runs the halo-conv ablations on the GPU box: tools/micro/c64_ablate.sh > gpurun_out/c64_ablate.log
set -e
for a in ${ABLS:-0 1 2 3 4 8 7 15}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iself-supervised-anomaly-detection_amd/csrc -DC64_ABL=$a tools/micro/c64_ablate.hip -o /tmp/c64_abl_$a 2>/dev/null
  timeout -k 5 60 /tmp/c64_abl_$a
done
