#!/usr/bin/env python3
"""Writes / extends tests/golden/oracle_fullsize.json: size + SHA-256 (+ factor statistics) of the ORACLE's stream for the full-size
configurations of BASELINE.json, so that the GPU tests and bench.py can compare the device stream byte for byte (by hash) at sizes
where the oracle takes minutes and tens of gigabytes of host memory.

    python tests/make_fullsize_golden.py english_2e9      # the metric's configuration: 2*10^9 B English, huff, t=2, flatten=1
    python tests/make_fullsize_golden.py english_256MiB   # configs[1]
    python tests/make_fullsize_golden.py dna_1e9_arith    # configs[2]: 10^9 B DNA, arithmetic coder, t=5
    python tests/make_fullsize_golden.py lz78_1e9         # configs[3]: 10^9 B English, lz78 + gamma

One entry per run (the oracle is single-threaded: ~6.5 min and ~50 GB for english_2e9); a progress line goes to stdout every 30 s.
The oracle is test infrastructure -- nothing here touches the GPU or the product library's compute paths (the text generators are
host code of the binding)."""
import hashlib
import json
import os
import resource
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tudocomp_amd as T
from oracle import oracle as O

PATH = os.path.join(ROOT, "tests", "golden", "oracle_fullsize.json")

CONFIGS = {
    # name: (generator, N, seed, compressor, coder, threshold, flatten)
    "english_2e9": ("english", 2 * 10**9, 42, "lcpcomp", "huff", 2, 1),
    "english_256MiB": ("english", 1 << 28, 42, "lcpcomp", "huff", 2, 1),
    "english_64MiB": ("english", 1 << 26, 42, "lcpcomp", "huff", 2, 1),
    "dna_1e9_arith": ("dna", 10**9, 7, "lcpcomp", "arithmetic", 5, 1),
    "lz78_1e9": ("english", 10**9, 42, "lz78", "gamma", 0, 0),
}


def run(name):
    gen, N, seed, comp, coder, thr, flat = CONFIGS[name]
    t0 = time.time()
    data = T.gen_english(N, seed) if gen == "english" else T.gen_dna(N, seed)
    print("%s: text generated in %.1f s" % (name, time.time() - t0), flush=True)
    entry = {"generator": gen, "N": N, "seed": seed, "compressor": comp, "coder": coder, "threshold": thr, "flatten": flat,
             "text_sha256": hashlib.sha256(data).hexdigest()}
    stop = threading.Event()

    def ticker():
        while not stop.wait(30):
            print("  ... oracle running, %.0f s, peak RSS %.1f GB" %
                  (time.time() - t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6), flush=True)
    th = threading.Thread(target=ticker, daemon=True)
    th.start()
    t1 = time.time()
    if comp == "lz78":
        out = O.lz78_gamma_compress(data)
        st = {}
    else:
        text = np.concatenate([data, np.zeros(1, dtype=np.uint8)])     # the generators emit no 0x00 / 0xFF: escaping = sentinel
        del data
        if coder == "huff":
            out, st = O.lcpcomp_huff_compress(text, thr, flat)
        else:
            out, st = O.lcpcomp_arith_compress(text, thr, flat)
    stop.set()
    entry.update({"size": len(out), "sha256": hashlib.sha256(out).hexdigest(), "oracle_seconds": round(time.time() - t1, 1),
                  "peak_rss_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1)})
    for k in ("factors", "num_flattened", "max_depth_lb", "maxlcp"):
        if k in st:
            entry[k] = int(st[k])
    return entry


if __name__ == "__main__":
    names = sys.argv[1:] or ["english_256MiB"]
    for name in names:
        e = run(name)
        allv = json.load(open(PATH)) if os.path.exists(PATH) else {}
        allv[name] = e
        with open(PATH, "w") as f:
            json.dump(allv, f, indent=1, sort_keys=True)
        print(name, json.dumps(e), flush=True)
