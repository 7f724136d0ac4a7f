#!/bin/bash
# Development aid: the per-process fast / slow mode of the ISA / flatten stages against the arena's address: N bench processes
N=${1:-6}
for i in $(seq 1 $N); do
  TDC_GPU_DEBUG_KNOBS=1 TDC_GPU_ARENA_LOG=1 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra 2> gpurun_out/mode_err.txt | python3 -c '
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=j["stages_ms"]
print("phi %.2f flatten %.2f factorize %.2f sa %.2f total %.2f" % (s["phi"], s["flatten"], s["factorize"], s["sa"], s["total"]))'
  grep arena gpurun_out/mode_err.txt | tail -2
done
