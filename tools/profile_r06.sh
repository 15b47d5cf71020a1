#!/bin/bash
# Round-6 evidence, part $1 (1 or 2) -- each part fits one gpurun call:
#   1: rocprofv3 kernel stats of the bench command + PMC traffic of the scoring kernel + SQ counters of the training kernels
#   2: per-dispatch traces of one replayed step (fp32 batch 256 / 32, precision 16 batch 256), precision-16 stats + PMC traffic
set -e
R=$PWD
if [ "$1" = "1" ]; then
  bash tools/profile_round.sh r06
else
  TAG=r06 bash tools/trace_step.sh 256 && TAG=r06 bash tools/trace_step.sh 32 && cd $R && \
  BENCH_ARGS="--train-precision 16" TAG=r06_p16 bash tools/trace_step.sh 256 && cd $R && bash tools/profile_p16.sh r06
fi
