"""Full-size parity check of BASELINE.json configs[2] and configs[3] (run on a GPU box; several minutes of CPU time for
the oracle).  Usage: python tests/gpu_config_check.py arith|lz78 [N]"""
import sys, time, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tudocomp_amd as T
from oracle import oracle as O

which = sys.argv[1]
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10**9
if which == "arith":
    data = T.gen_dna(N, 7)
    text = np.concatenate([data, np.zeros(1, dtype=np.uint8)])
    with T.Context(0) as ctx:
        t0 = time.time()
        got, st = ctx.lcpcomp_compress(text, 5, 1, T.CODER_ARITH)
        print("GPU lcpcomp(coder=arithmetic,threshold=5) on %d B DNA: %.2f s wall, device %.1f ms, %d bytes" % (N, time.time() - t0, st["ms_total"], len(got)), flush=True)
        print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in st.items() if k.startswith("ms_") or k in ("factors", "levels", "small_levels")}, flush=True)
    t0 = time.time()
    want, _ = O.lcpcomp_arith_compress(text, 5, 1)
    print("oracle: %.1f s, %d bytes" % (time.time() - t0, len(want)), flush=True)
else:
    data = T.gen_english(N, 42)
    with T.Context(0) as ctx:
        t0 = time.time()
        got, st = ctx.lz78_compress(data)
        print("GPU lz78(coder=gamma) on %d B english: %.2f s wall (host parse + device pack %.1f ms), %d phrases, %d bytes" % (N, time.time() - t0, st["ms_total"], st["factors"], len(got)), flush=True)
    t0 = time.time()
    want = O.lz78_gamma_compress(data)
    print("oracle: %.1f s, %d bytes" % (time.time() - t0, len(want)), flush=True)
ok = len(got) == len(want) and hashlib.sha256(got).digest() == hashlib.sha256(want).digest()
print("bit-exact:", ok, flush=True)
sys.exit(0 if ok else 1)
