#!/bin/bash
# Development aid: stage lengths in the KERNEL TRACE of one step (bench.py --steps 1 --warmup 1), for a list of option settings -- stable to
# ~0.1 ms where an A/B of whole steps wanders by a millisecond:  tools/stage_spans.sh "VAR=x" "VAR=y VAR2=z" ...
#   up+sa = first kernel of the step (the histogram of the first chunk) -> fused scatter; sa = last kernel behind the upload -> fused scatter; leaf = last partition scatter -> first pass over the head flags; rounds = that -> fused
#   scatter; phi = fused scatter -> candidates; fact = candidates -> flatten; flat = flatten -> pack
R=$PWD
for cfg in "$@"; do
  OUT=$R/gpurun_out/stagespan; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && env TDC_GPU_DEBUG_KNOBS=1 $cfg timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/err.txt )
  python3 - "$cfg" $OUT <<'PY'
import csv, glob, sys
cfg, out = sys.argv[1:3]
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)
if not f: print(cfg, "no trace"); sys.exit(0)
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f[0]))))
def last(pat, hi=None): return [i for i, r in enumerate(rows[:hi]) if pat in r[2]][-1]
def first(pat, lo): return next(i for i in range(lo, len(rows)) if pat in rows[i][2])
try:
    i0 = last("ws_scatter_kernel<2, false, true")
    up = last("byte_hist", i0)
    up0 = up
    while up0 > 0 and rows[up0][0] - rows[up0 - 1][0] < 20e6 : up0 -= 1      # first kernel of the step (a gap of 20 ms and more: the step before)
    j = first("sa_flag_count", i0); k = first("fs_count", j); c = first("cand_class", k); fl = first("flatten_init", c); pk = first("pack_cls", fl)
    ms = lambda a, b: (b - a) / 1e6
    print("%-44s up+sa %.2f | sa %.2f (leaf %.2f rounds %.2f) phi %.2f fact %.2f flat %.2f | sum %.2f" % (cfg, ms(rows[up0][0], rows[k][0]), ms(rows[up][1], rows[k][0]), ms(rows[i0][1], rows[j][0]), ms(rows[j][0], rows[k][0]),
          ms(rows[k][0], rows[c][0]), ms(rows[c][0], rows[fl][0]), ms(rows[fl][0], rows[pk][0]), ms(rows[up][1], rows[pk][0])))
except Exception as e: print(cfg, "trace not understood:", e)
PY
  rm -rf $OUT
done
