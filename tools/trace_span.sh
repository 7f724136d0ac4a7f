#!/bin/bash
# Development aid: every dispatch between the first kernel matching FROM and the next kernel matching TO (second occurrence = the timed
# step of `bench.py --steps 1 --warmup 1`), with start offsets, durations and the idle time in front of each
#   tools/trace_span.sh FROM TO TAG
FROM=$1; TO=$2; TAG=${3:-x}
R=$PWD
OUT=$R/gpurun_out/span_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/err.txt
cd $R
python3 - "$FROM" "$TO" $OUT > gpurun_out/span_$TAG.txt <<'PY'
import csv, glob, sys, re
frm, to, out = sys.argv[1:4]
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
starts = [i for i, r in enumerate(rows) if re.search(frm, r[2])]
# group consecutive matches: occurrences separated by > 100 ms are different steps
occ = []
for i in starts:
    if not occ or rows[i][0] - rows[occ[-1]][0] > 100e6: occ.append(i)
i0 = occ[1] if len(occ) > 1 else occ[0]
t0 = rows[i0][0]; last_end = t0
for s, e, n in rows[i0:]:
    print("%9.3f ms  idle %7.1f us  +%8.3f ms  %s" % ((s - t0) / 1e6, max(0, s - last_end) / 1e3, (e - s) / 1e6, n[:110]))
    last_end = max(last_end, e)
    if re.search(to, n) and s > t0: break
PY
rm -rf $OUT
cat gpurun_out/span_$TAG.txt | cut -c1-170 | head -150
