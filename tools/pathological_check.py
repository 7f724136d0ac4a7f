"""Texts with extreme repeat structure (runs, periodic texts, Fibonacci words) through both strategies: round trip and time.
Run on a GPU box: python tools/pathological_check.py [N]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tudocomp_amd as T
from oracle import oracle as O

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
a, b = b"a", b"ab"
while len(b) < N:
    a, b = b, b + a
import numpy as np
rng = np.random.default_rng(5)
base = rng.integers(97, 123, N // 20, dtype=np.uint8)
versions = []
for _ in range(20):                                   # a versioned collection: 20 copies of one document with 0.2 % point edits each
    v = base.copy()
    idx = rng.integers(0, len(v), max(1, len(v) // 500))
    v[idx] = rng.integers(97, 123, len(idx), dtype=np.uint8)
    versions.append(v.tobytes())
    base = v
runs = b"".join(bytes([int(c)]) * int(l) for c, l in zip(rng.integers(97, 100, N // 50), rng.integers(1, 100, N // 50)))
cases = {"a^N": b"a" * N, "(ab)^N/2": b"ab" * (N // 2), "(abc)^k x (abc)^k": b"abc" * (N // 6) + b"x" + b"abc" * (N // 6), "fibonacci": b[:N],
         "20 versions": b"".join(versions), "random runs": runs[:N]}
with T.Context(0) as ctx:
    for name, data in cases.items():
        text = O.escape(data)
        for comp, cn in ((T.COMP_ARRAYS, "arrays"), (T.COMP_MAXLCP, "max_lcp")):
            t0 = time.time()
            out, st = ctx.lcpcomp_compress(text, 5, 1, T.CODER_HUFF, comp)
            dt = time.time() - t0
            want = (O.lcpcomp_huff_compress if comp == T.COMP_ARRAYS else O.lcpcomp_maxlcp_huff_compress)(text, 5, 1)[0]
            print("%-20s %-8s %.3f s  factors %d levels %d probes %s  == oracle: %s" % (name, cn, dt, st["factors"], st["levels"], st.get("probes"), out == want), flush=True)
