#!/bin/bash
# Collect the evidence behind bench.py's line on a GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh TAG      -> gpurun_out/prof_TAG/{bench.json, stats/, pmc_fetch/, pmc_write/}
# Kernel trace + stats and the two PMC passes are separate rocprofv3 runs (counters are never combined with tracing
# domains other than --kernel-trace).
set -u
TAG=${1:-x}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $OUT/stats_bench.json 2> $OUT/stats.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc_write.err
cd $R
# keep only the small summaries (the merged directory is capped at 64 MiB)
find $OUT -name "*kernel_trace.csv" -path "*stats*" -delete
ls -R $OUT | head -40
