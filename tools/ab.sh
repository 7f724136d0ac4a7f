#!/bin/bash
# A/B runs of bench.py under different env toggles: tools/ab.sh "VAR1=a VAR2=b" "VAR1=c" ...
# prints value, stage times and the top kernel classes for every setting
for cfg in "$@"; do
  echo "== $cfg"
  env TDC_GPU_DEBUG_KNOBS=1 $cfg python bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} 2>>gpurun_out/ab_stderr.log | python3 -c '
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print("value",j["value"],"ms",j["ms_per_step"])
print("stages",j["stages_ms"])
k=j["kernels"]
print(" ".join("%s=%.2f"%(n,v["ms_per_step"]) for n,v in sorted(k.items(), key=lambda x:-x[1]["ms_per_step"])))
print("roof",j["roofline"]["kernel"],j["roofline"]["achieved"])
'
done
