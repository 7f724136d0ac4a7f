#!/bin/bash
# Development aid: idle gaps of the device inside one bench step (kernel trace of all kernels; gaps >= MIN_US between the end of one
# dispatch and the start of the next on any queue are listed with the kernels on both sides)
#   tools/trace_gaps.sh TAG [MIN_US]   -> gpurun_out/gaps_TAG.txt
TAG=${1:-x}; MIN=${2:-150}
R=$PWD
OUT=$R/gpurun_out/gaps_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/err.txt
cd $R
python3 - $OUT $MIN > gpurun_out/gaps_$TAG.txt <<'PY'
import csv, glob, sys
out, mn = sys.argv[1], float(sys.argv[2])
ev = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
# runs of the pipeline: a run ends with its last download; label every gap with the number of window kernels seen so far
t0 = ev[0][0]
busy_end = ev[0][0]; prev = None; tot = {}; run = 0
for s, e, n in ev:
    if "window_eager_kernel" in n or "window_levels_kernel" in n: run += 1
    if s > busy_end:
        g = (s - busy_end) / 1e3
        if g >= mn and g < 50000:
            print("run %d  %9.3f ms: idle %7.1f us   after %-60s before %s" % (run, (busy_end - t0) / 1e6, g, prev, n))
        if g < 50000: tot[run] = tot.get(run, 0) + g
    if e > busy_end: busy_end = e; prev = n
print("idle per run (ms; a run is counted from its window kernel to the next one's):", {k: round(v / 1e3, 2) for k, v in tot.items()})
PY
rm -rf $OUT
tail -70 gpurun_out/gaps_$TAG.txt
