"""Randomised check of the star step (suffix_array.hip sa_star_*) on a GPU box: copy-structured texts of 2^25 .. 2^25 + 8 M bytes with random
alphabets, block lengths, copy rates, copies of copies and overlapping sources; the suffix array with the step must equal the one
without it (the pair step + doubling rounds, validated against the oracle since round 5), and every n-th text is also compared with
the oracle's suffix array.  Usage: python3 tools/star_stress.py [seconds] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tudocomp_amd as T
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def make(n):
    sigma = int(rng.choice([2, 3, 4, 4, 6, 12]))
    blk = int(rng.choice([64, 300, 1000, 4096, 4096, 20000, 70000]))
    rate = float(rng.choice([0.1, 0.25, 0.5]))
    recent = bool(rng.integers(0, 2))              # sources close to the copy (overlapping windows, copies of copies)
    mutate = float(rng.choice([0.0, 0.0, 0.001]))
    out = np.empty(n + blk, dtype=np.uint8)
    pos = 0
    while pos < n:
        if pos >= blk and rng.random() < rate:
            lo = max(0, pos - 50 * blk) if recent else 0
            src = int(rng.integers(lo, pos - blk + 1))
            out[pos:pos + blk] = out[src:src + blk]
            if mutate:
                k = rng.random(blk) < mutate
                out[pos:pos + blk][k] = rng.integers(0, sigma, int(k.sum()), dtype=np.uint8) + 65
        else:
            out[pos:pos + blk] = rng.integers(0, sigma, blk, dtype=np.uint8) + 65
        pos += blk
    return out[:n].tobytes(), (sigma, blk, rate, recent, mutate)


t0 = time.time()
cnt = taken = 0
with T.Context(0) as star, T.Context(0, options={"sa_stars": 0}) as plain:
    while time.time() - t0 < budget:
        n = (1 << 25) + int(rng.integers(1, 8 << 20))
        data, par = make(n)
        text = O.escape(data)
        a, st = star.lcpcomp_compress(text, threshold=3, flatten=1)
        b, _ = plain.lcpcomp_compress(text, threshold=3, flatten=1)
        sa1, isa1 = star.suffix_array(text)
        sa2, _ = plain.suffix_array(text)
        ok = a == b and np.array_equal(sa1, sa2) and np.array_equal(isa1[sa1], np.arange(len(sa1), dtype=np.uint32))
        if ok and cnt % 8 == 0:
            ok = np.array_equal(sa1, O.suffix_array(text))
        cnt += 1
        taken += 1 if st["sa_star_chains"] else 0
        if not ok:
            print("MISMATCH", par, n, "streams equal:", a == b, "sa equal:", bool(np.array_equal(sa1, sa2)), {k: st[k] for k in ("maxlcp", "sa_star_chains", "sa_rounds")})
            sys.exit(1)
        if cnt % 10 == 0:
            print("... %d texts (%d took the star step), %.0f s" % (cnt, taken, time.time() - t0), flush=True)
print("star stress ok: %d texts (%d took the star step) in %.0f s" % (cnt, taken, time.time() - t0))
