#!/bin/bash
# Development aid: rocprofv3 kernel stats of tools/run_once.py ARGS...   -> gpurun_out/stats_TAG.txt (top 40 kernels, both calls summed)
#   tools/stats_once.sh TAG dna 1000000000 5 arith
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/stats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/run_once.py "$@" > $OUT/run.txt 2> $OUT/err.txt
cd $R
python3 - $OUT > gpurun_out/stats_$TAG.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(open(sys.argv[1] + "/run.txt").read().strip())
for r in rows[:40]:
    print("%9.3f ms %6s calls  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], r["Name"][:110]))
PY
rm -rf $OUT
cat gpurun_out/stats_$TAG.txt
