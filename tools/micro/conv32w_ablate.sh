#!/bin/bash
# float instantiation of the register-fed conv under the ablation bits: 1 = no filter loads, 2 = no halo loads, 4 = no epilogue, 16 = no A reads
set -e
for a in ${ABLS:-0 1 2 4 7 23}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iself-supervised-anomaly-detection_amd/csrc -DCONV16W_ABL=$a ${EXTRA} tools/micro/conv32w_ablate.hip -o /tmp/conv32w_abl_$a 2>/dev/null
  timeout -k 5 60 /tmp/conv32w_abl_$a
done
