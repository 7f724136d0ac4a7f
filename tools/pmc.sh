#!/bin/bash
# Development aid: SQ / TCC counters per kernel of one compression (tools/run_once.py) on a GPU box.
#   tools/pmc.sh TAG "COUNTER1 COUNTER2 ..." english 268435456 2   -> gpurun_out/pmc_TAG.txt
set -u
TAG=${1:-x}; CNT=${2:-"SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"}
GEN=${3:-english}; N=${4:-268435456}; THR=${5:-2}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
timeout 900 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d /tmp/pmc_$TAG -- python3 $R/tools/run_once.py $GEN $N $THR > $R/gpurun_out/pmc_$TAG.run 2>&1
python3 - /tmp/pmc_$TAG $R/gpurun_out/pmc_$TAG.txt <<'PY'
import collections, csv, glob, sys
src, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
names = []
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("tdc::", "")[:60]
        c = r["Counter_Name"]
        if c not in names: names.append(c)
        acc[k][c] += float(r["Counter_Value"])
        if c == names[0]: cnt[k] += 1
with open(out, "w") as g:
    g.write("%-60s %6s " % ("kernel (both calls summed)", "n") + " ".join("%16s" % n[-16:] for n in names) + "\n")
    for k in sorted(acc, key=lambda k: -acc[k][names[0]])[:45]:
        g.write("%-60s %6d " % (k, cnt[k]) + " ".join("%16.4g" % acc[k][n] for n in names) + "\n")
print(open(out).read())
PY
cd $R
