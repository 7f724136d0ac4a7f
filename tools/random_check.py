"""Incompressible and low-entropy inputs (random bytes, random bits as '0'/'1', random DNA): time and round trip.
Run on a GPU box: python tools/random_check.py [N]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tudocomp_amd as T
from oracle import oracle as O

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 64 << 20
rng = np.random.default_rng(9)
cases = {"random bytes": rng.integers(0, 256, N, dtype=np.uint8).tobytes(),
         "random bits": rng.integers(48, 50, N, dtype=np.uint8).tobytes(),
         "random acgt": np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, N)].tobytes()}
with T.Context(0) as ctx:
    for name, data in cases.items():
        text = O.escape(data)
        for thr in (2, 5):
            for _ in range(2):
                t0 = time.time()
                out, st = ctx.lcpcomp_compress(text, thr, 1)
                dt = time.time() - t0
            back, _ = ctx.lcpcomp_decompress(out)
            print("%-13s t=%d n=%d: %.3f s wall, device %.1f ms (sa %.1f plcp %.1f fact %.1f flat %.1f enc %.1f) window_pass %s out/in %.3f roundtrip %s"
                  % (name, thr, len(text), dt, st["ms_total"], st["ms_sa"], st["ms_plcp"], st["ms_factorize"], st["ms_flatten"], st["ms_encode"],
                     st.get("window_pass"), len(out) / len(data), back == text), flush=True)
