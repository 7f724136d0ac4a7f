"""Randomised check of the eager level runs (factorize_eager.hip) and of the pair step of the suffix array on a GPU box: texts made of
copied blocks (random block size, alphabet, mutation rate, copies of copies), long runs and periodic stretches mixed with random
background -- hundreds to thousands of LCP levels each --, every stream compared byte for byte with the oracle's, eager runs on and off.
Usage: python3 tools/eager_stress.py [seconds] [seed]"""
import os; os.environ.setdefault("TDC_GPU_DEBUG_KNOBS", "1")   # (development tool: the TDC_GPU_* variables below are applied -- include/tdc_gpu.h, options)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tudocomp_amd as T
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def make(n):
    sigma = int(rng.integers(2, 24))
    block = int(rng.integers(150, 9000))
    pcopy = float(rng.uniform(0.15, 0.6))
    mutate = float(rng.choice([0.0, 0.0, 0.0005, 0.003, 0.02]))
    out = np.zeros(n, dtype=np.uint8)
    ln = 0
    while ln < n:
        m = min(int(block * rng.uniform(0.5, 1.5)), n - ln)
        kind = rng.random()
        if ln >= m and kind < pcopy:
            src = int(rng.integers(0, ln - m + 1))
            out[ln:ln + m] = out[src:src + m]
            if mutate:
                k = rng.random(m) < mutate
                out[ln:ln + m][k] = rng.integers(65, 65 + sigma, int(k.sum()), dtype=np.uint8)
        elif kind > 0.97:
            out[ln:ln + m] = 65 + int(rng.integers(0, sigma))                   # a run
        elif kind > 0.94:
            u = rng.integers(65, 65 + sigma, int(rng.integers(2, 12)), dtype=np.uint8)
            out[ln:ln + m] = np.resize(u, m)                                     # a periodic stretch
        else:
            out[ln:ln + m] = rng.integers(65, 65 + sigma, m, dtype=np.uint8)
        ln += m
    return out.tobytes()


t0 = time.time()
cases = eager = levels = 0
os.environ["TDC_GPU_EAGER"] = "1"
on = T.Context(0)
os.environ["TDC_GPU_EAGER"] = "0"
off = T.Context(0)
try:
    while time.time() - t0 < budget:
        n = int(rng.integers(200_000, 4_000_000))
        thr = int(rng.choice([1, 2, 2, 3, 5, 8]))
        fl = int(rng.integers(0, 2))
        text = O.escape(make(n))
        want, _ = O.lcpcomp_huff_compress(text, thr, fl)
        got, st = on.lcpcomp_compress(text, threshold=thr, flatten=fl)
        got0, st0 = off.lcpcomp_compress(text, threshold=thr, flatten=fl)
        if got != want or got0 != want:
            print("MISMATCH n %d thr %d flatten %d: eager %s lazy %s (phases %d, levels %d / %d, sa_mode %d)" % (
                len(text), thr, fl, got == want, got0 == want, st["eager_phases"], st["eager_levels"], st["levels"], st["sa_mode"]))
            open("gpurun_out/eager_stress_fail.bin", "wb").write(text)
            sys.exit(1)
        cases += 1
        if cases % 25 == 0: print("... %d texts, %.0f s" % (cases, time.time() - t0), flush=True)
        eager += 1 if st["eager_phases"] else 0
        levels += st["eager_levels"]
finally:
    on.close(); off.close()
print("eager stress ok: %d texts in %.0f s (%d with eager phases, %d levels inside them), all streams equal to the oracle's" % (cases, time.time() - t0, eager, levels))
