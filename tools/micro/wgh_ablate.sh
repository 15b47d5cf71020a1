#!/bin/bash
set -e
for a in ${ABLS:-0 1 2 3}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iself-supervised-anomaly-detection_amd/csrc -DWGH_ABL=$a tools/micro/wgh_ablate.hip -o /tmp/wgh_abl_$a 2>/dev/null
  timeout -k 5 60 /tmp/wgh_abl_$a
done
