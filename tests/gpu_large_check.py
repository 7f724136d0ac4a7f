"""Ad-hoc large-size check (run on a GPU box): compress a big synthetic text on the GPU, decode it with the oracle's
decoder, compare.  Usage: python tests/gpu_large_check.py english|dna N [threshold] [lcpcomp|lzss_lcp|max_lcp|sle] [exact]
(exact: also compare with the oracle's compressor output byte by byte -- minutes of CPU time per 100 MB)"""
import sys, time, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tudocomp_amd as T
from oracle import oracle as O

gen, N = sys.argv[1], int(float(sys.argv[2]))
thr = int(sys.argv[3]) if len(sys.argv) > 3 else (2 if gen == "english" else 5)
algo = sys.argv[4] if len(sys.argv) > 4 else "lcpcomp"
t0 = time.time()
data = T.gen_english(N, 42) if gen == "english" else T.gen_dna(N, 7)
text = np.concatenate([data, np.zeros(1, dtype=np.uint8)])
print("generated %d bytes in %.1f s" % (N, time.time() - t0), flush=True)
with T.Context(0) as ctx:
    t0 = time.time()
    if algo == "lcpcomp": out, st = ctx.lcpcomp_compress(text, thr, 1)
    elif algo == "max_lcp": out, st = ctx.lcpcomp_compress(text, thr, 1, T.CODER_HUFF, T.COMP_MAXLCP)
    elif algo == "sle": out, st = ctx.lcpcomp_compress(text, thr, 1, T.CODER_SLE)
    else: out, st = ctx.lzss_lcp_compress(text, thr)
    print("compressed in %.2f s wall; device %.1f ms; out %d (ratio %.4f)" % (time.time() - t0, st["ms_total"], len(out), len(out) / N), flush=True)
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()}, flush=True)
t0 = time.time()
back = O.lcpcomp_sle_decompress(out, 3) if algo == "sle" else O.lcpcomp_huff_decompress(out)
ok = len(back) == N + 1 and hashlib.sha256(back).digest() == hashlib.sha256(text.tobytes()).digest()
print("oracle decode in %.1f s: roundtrip %s" % (time.time() - t0, "OK" if ok else "MISMATCH"), flush=True)
if ok and len(sys.argv) > 5 and sys.argv[5] == "exact":
    t0 = time.time()
    fn = {"lcpcomp": O.lcpcomp_huff_compress, "max_lcp": O.lcpcomp_maxlcp_huff_compress, "sle": O.lcpcomp_sle_compress}[algo]
    want, _ = fn(text.tobytes(), thr, 1)
    ok = want == out
    print("oracle compress in %.1f s: %s" % (time.time() - t0, "IDENTICAL" if ok else "DIFFERENT"), flush=True)
sys.exit(0 if ok else 1)
