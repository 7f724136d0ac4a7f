"""Randomised end-to-end check on a GPU box: many structured random texts (sizes 70 K .. 3 M, so that the suffix-array refinement, the window
pass with its small halo, the one-workgroup level pipeline and the purge policy all run), every stream compared byte for byte with the
oracle's.  Usage: python3 tools/stress_check.py [seconds] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tudocomp_amd as T
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def make(kind, n):
    if kind == 0:                                              # words from a small vocabulary
        v = int(rng.integers(8, 400))
        voc = [bytes(rng.integers(97, 97 + int(rng.integers(2, 26)), int(rng.integers(1, 9)), dtype=np.uint8)) for _ in range(v)]
        idx = (rng.zipf(1.3, n // 4) - 1) % v
        return b" ".join(voc[int(i)] for i in idx)[:n]
    if kind == 1:                                              # random background with copies of earlier windows (long repeats)
        sig = int(rng.integers(2, 6))
        out = bytearray(rng.integers(65, 65 + sig, n, dtype=np.uint8).tobytes())
        for _ in range(int(rng.integers(1, 40))):
            ln = int(rng.integers(10, min(n // 3, 20000)))
            src = int(rng.integers(0, n - ln)); dst = int(rng.integers(0, n - ln))
            out[dst:dst + ln] = out[src:src + ln]
        return bytes(out)
    if kind == 2:                                              # periodic with mutations
        unit = bytes(rng.integers(97, 101, int(rng.integers(2, 300)), dtype=np.uint8))
        a = np.frombuffer((unit * (n // len(unit) + 1))[:n], dtype=np.uint8).copy()
        k = int(rng.integers(0, 50))
        a[rng.integers(0, n, k)] = 122
        return a.tobytes()
    if kind == 3:
        return T.gen_english(n, int(rng.integers(0, 1 << 30))).tobytes()
    if kind == 4:
        return T.gen_dna(n, int(rng.integers(0, 1 << 30))).tobytes()
    return bytes(rng.integers(1, 255, n, dtype=np.uint8))     # near-incompressible bytes (escaping exercised by 0xFF never: range 1..254)


t0 = time.time()
cases = 0
with T.Context(0) as ctx:
    while time.time() - t0 < budget:
        kind = int(rng.integers(0, 6))
        n = int(rng.integers(70_000, 3_000_000 if kind in (3, 4, 5) else 1_200_000))
        thr = int(rng.choice([1, 2, 2, 3, 5, 8]))
        fl = int(rng.integers(0, 2))
        text = O.escape(make(kind, n))
        want, _ = O.lcpcomp_huff_compress(text, thr, fl)
        got, st = ctx.lcpcomp_compress(text, threshold=thr, flatten=fl)
        if got != want:
            print("MISMATCH kind %d n %d thr %d flatten %d (window_pass %d, levels %d)" % (kind, len(text), thr, fl, st["window_pass"], st["levels"]))
            sys.exit(1)
        cases += 1
        if cases % 200 == 0: print("... %d texts, %.0f s" % (cases, time.time() - t0), flush=True)
print("stress ok: %d texts in %.0f s, all streams equal to the oracle's" % (cases, time.time() - t0))
