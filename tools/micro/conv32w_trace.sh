#!/bin/bash
# in-kernel time stamps of the register-fed conv under build variants: VARIANTS="name:flags ..." (flags joined by commas)
set -e
for v in ${VARIANTS:-base:}; do
  name=${v%%:*}; flags=$(echo "${v#*:}" | tr ',' ' ')
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iself-supervised-anomaly-detection_amd/csrc -DCONV16W_TRACE=17 $flags tools/micro/conv32w_trace.hip -o /tmp/conv32w_trace_$name 2>/dev/null
  echo "#### $name ($flags)"
  timeout -k 5 120 /tmp/conv32w_trace_$name
done
