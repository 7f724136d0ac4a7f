"""Summarise a rocprofv3 --kernel-trace directory of tools/run_once.py (two identical calls): per-kernel totals and the launch
sequence of the SECOND call.  Usage: kernel_times.py TRACE_DIR OUT_PREFIX"""
import collections
import csv
import glob
import sys

src, out = sys.argv[1], sys.argv[2]
f = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]                      # the second of the two identical calls
short = lambda n: n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("tdc::", "")[:70]
tot = collections.OrderedDict()
with open(out + "_seq.txt", "w") as g:
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        n = short(r["Kernel_Name"])
        a = tot.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += d
        g.write("%9.3f %8.3f %s grid=%s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e6, d, n, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
with open(out + ".txt", "w") as g:
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
    g.write("span %.3f ms, kernel time %.3f ms, %d launches\n" % (span, sum(v[1] for v in tot.values()), len(rows)))
    for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        g.write("%9.3f ms %5d  %s\n" % (d, c, n))
print(open(out + ".txt").read())
