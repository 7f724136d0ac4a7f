"""A long chain of versions (each a copy of the previous one with a few point edits): deep dependency chains for flatten and
pointer jumping.  Run on a GPU box: python tools/version_chain_check.py [versions] [length]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tudocomp_amd as T
from oracle import oracle as O

V = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
Ln = int(sys.argv[2]) if len(sys.argv) > 2 else 500
rng = np.random.default_rng(3)
v = rng.integers(97, 123, Ln, dtype=np.uint8)
parts = []
for _ in range(V):
    parts.append(v.tobytes())
    v = v.copy()
    v[rng.integers(0, Ln, 2)] = rng.integers(97, 123, 2, dtype=np.uint8)
text = O.escape(b"".join(parts))
with T.Context(0) as ctx:
    for thr in (2, 5):
        for comp, name, fn in ((T.COMP_ARRAYS, "arrays", O.lcpcomp_huff_compress), (T.COMP_MAXLCP, "max_lcp", O.lcpcomp_maxlcp_huff_compress)):
            for fl in (1, 0):
                t0 = time.time()
                out, st = ctx.lcpcomp_compress(text, thr, fl, T.CODER_HUFF, comp)
                dt = time.time() - t0
                want, wst = fn(text, thr, fl)
                back, ds = ctx.lcpcomp_decompress(out)
                print("%d versions x %d, t=%d %s flatten=%d: %.3f s, factors %d, flatten rounds %s (max_depth_lb %s), decode rounds %d, == oracle %s, roundtrip %s"
                      % (V, Ln, thr, name, fl, dt, st["factors"], st.get("flatten_rounds"), st.get("max_depth_lb"), ds["rounds"], out == want, back == text), flush=True)
