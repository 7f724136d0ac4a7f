#!/bin/bash
# Development aid: kernel trace of ONE compression (after one warm-up) on a GPU box.
#   tools/ktrace.sh english 268435456 2 [TAG]   -> gpurun_out/ktrace_TAG.txt (per-kernel totals of the last call) + _seq.txt
set -u
GEN=${1:-english}; N=${2:-268435456}; THR=${3:-2}; TAG=${4:-x}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$TAG
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$TAG -- python3 $R/tools/run_once.py $GEN $N $THR ${5:-huff} > $R/gpurun_out/ktrace_$TAG.run 2>&1
python3 $R/tools/kernel_times.py /tmp/kt_$TAG $R/gpurun_out/ktrace_$TAG
cd $R
