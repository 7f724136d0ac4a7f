#!/bin/bash
# Development aid: tools/trace_kernel.sh for several env settings: tools/trace_env.sh PATTERN "A=1" "A=2 B=3" ...
PAT=$1; shift
i=0
for cfg in "$@"; do
  i=$((i+1))
  echo "== $cfg"
  env TDC_GPU_DEBUG_KNOBS=1 $cfg TDC_GPU_LEVEL_LOG=1 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra 2>&1 >/dev/null | grep "flatten round" | head -12
  env TDC_GPU_DEBUG_KNOBS=1 $cfg tools/trace_kernel.sh "$PAT" env$i | awk 'NR<=14'
done
