"""Randomised check of the wide suffix sort (wsort.hip: partition levels, leaf stage with fill bits / peel, run kernels, text rounds, fall-backs)
on a GPU box: texts of 2 .. 24 MB of many kinds, the device's suffix array against the oracle's (SA-IS).  Every text is written to
stdout as it is done.  Usage: python3 tools/wsort_stress.py [seconds] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tudocomp_amd as T
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def make(n):
    kind = int(rng.integers(0, 8))
    if kind == 0: return "english", T.gen_english(n, int(rng.integers(1, 1 << 30))).tobytes()
    if kind == 1: return "dna", T.gen_dna(n, int(rng.integers(1, 1 << 30))).tobytes()
    if kind == 2:
        s = int(rng.choice([2, 3, 5, 17, 64, 200]))
        return "random sigma %d" % s, (rng.integers(0, s, n, dtype=np.uint8) + 1).tobytes()
    if kind == 3:                                              # words over a small vocabulary: heavy ties on the first key word
        v = [bytes(rng.integers(97, 123, int(rng.integers(1, 9)), dtype=np.uint8)) for _ in range(int(rng.choice([4, 30, 500])))]
        idx = rng.integers(0, len(v), n // 4)
        return "vocabulary %d" % len(v), b" ".join(v[i] for i in idx)[:n]
    if kind == 4:                                              # periodic with noise
        per = bytes(rng.integers(97, 101, int(rng.integers(1, 300)), dtype=np.uint8))
        a = np.frombuffer((per * (n // len(per) + 1))[:n], dtype=np.uint8).copy()
        k = rng.random(n) < float(rng.choice([0.0, 1e-5, 1e-3]))
        a[k] = rng.integers(97, 123, int(k.sum()), dtype=np.uint8)
        return "periodic %d" % len(per), a.tobytes()
    if kind == 5:                                              # runs
        vals = rng.integers(97, 100, n // 50 + 1, dtype=np.uint8)
        lens = rng.integers(1, 100, n // 50 + 1)
        return "runs", np.repeat(vals, lens)[:n].tobytes()
    if kind == 6:                                              # english with a planted long repeat
        d = bytearray(T.gen_english(n, int(rng.integers(1, 1 << 30))).tobytes())
        L = int(rng.integers(1000, n // 4))
        d[n - L:] = d[:L]
        return "english + repeat %d" % L, bytes(d)
    a = rng.integers(0, 2, n, dtype=np.uint8) * int(rng.integers(1, 255)) + 1
    return "two symbols", a.tobytes()


t0 = time.time()
cnt = 0
with T.Context(0) as ctx:
    while time.time() - t0 < budget:
        n = int(rng.integers(2 << 20, 24 << 20))
        name, data = make(n)
        text = O.escape(data)
        sa, isa = ctx.suffix_array(text)
        ok = np.array_equal(sa, O.suffix_array(text)) and np.array_equal(isa[sa], np.arange(len(sa), dtype=np.uint32))
        cnt += 1
        print("%4d %-24s n=%9d %s  %.0f s" % (cnt, name, len(text), "ok" if ok else "MISMATCH", time.time() - t0), flush=True)
        if not ok:
            sys.exit(1)
print("wsort stress ok: %d texts in %.0f s" % (cnt, time.time() - t0))
