"""Development aid: per-level figures of the eager level kernel (TDC_GPU_LEVEL_LOG) on a DNA text: histogram of cycles per level by
entries per level.  Usage: python3 tools/eager_levels_log.py [N]   (stderr of the library is parsed)"""
import os, sys, subprocess, re, collections
N = sys.argv[1] if len(sys.argv) > 1 else "268435456"
env = dict(os.environ, TDC_GPU_DEBUG_KNOBS="1", TDC_GPU_LEVEL_LOG="1", TDC_GPU_EAGER_DUMP="1")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = subprocess.run([sys.executable, os.path.join(root, "tools", "run_once.py"), "dna", N, "5", "arith"], env=env, capture_output=True, text=True)
rows = []
for l in p.stderr.splitlines():
    m = re.match(r"\s*eager level (\d+): entries (\d+) factors (\d+) cycles (\d+)", l)
    if m: rows.append(tuple(int(x) for x in m.groups()))
print(p.stdout.strip()[-300:])
print("levels logged:", len(rows))
if rows:
    tot = sum(r[3] for r in rows)
    print("total cycles %d (%.2f ms at 100 MHz counter)" % (tot, tot / 1e5))
    b = collections.defaultdict(lambda: [0, 0, 0])
    for L, e, f, cy in rows:
        k = 0 if e == 0 else (1 if e <= 4 else (2 if e <= 16 else (3 if e <= 64 else (4 if e <= 256 else 5))))
        b[k][0] += 1; b[k][1] += cy; b[k][2] += e
    names = ["0", "1-4", "5-16", "17-64", "65-256", ">256"]
    for k in sorted(b): print("entries %-7s levels %6d  cycles/level %8.0f  share %5.1f %%  avg entries %.1f" % (names[k], b[k][0], b[k][1] / b[k][0], 100.0 * b[k][1] / tot, b[k][2] / b[k][0]))
