#!/bin/bash
# Development aid: per-launch durations of the kernels whose name matches PATTERN in one bench step
#   tools/trace_kernel.sh PATTERN TAG   -> gpurun_out/trace_TAG.txt   (run through gpurun from the repo root)
PAT=$1; TAG=${2:-x}
R=$PWD
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/err.txt
cd $R
python3 - "$PAT" $OUT > gpurun_out/trace_$TAG.txt <<'PY'
import csv, glob, sys, re
pat, out = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if re.search(pat, r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.3f ms  +%8.3f ms  grid %9s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, r.get("Grid_Size", "?"), r["Kernel_Name"][:100]))
PY
rm -rf $OUT
cat gpurun_out/trace_$TAG.txt | tail -60
