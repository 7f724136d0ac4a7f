"""One compression of a synthetic text with a given coder / strategy (profiling aid).
Usage: python3 tools/run_once.py english|dna N threshold [huff|sle|ascii|arith] [arrays|max_lcp|plcppeaks]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tudocomp_amd as T

gen, N, thr = sys.argv[1], int(float(sys.argv[2])), int(sys.argv[3])
coder = {"huff": T.CODER_HUFF, "sle": T.CODER_SLE, "ascii": T.CODER_ASCII, "arith": T.CODER_ARITH}[sys.argv[4] if len(sys.argv) > 4 else "huff"]
comp = {"arrays": T.COMP_ARRAYS, "max_lcp": T.COMP_MAXLCP, "plcppeaks": T.COMP_PLCPPEAKS}[sys.argv[5] if len(sys.argv) > 5 else "arrays"]
data = T.gen_english(N, 42) if gen == "english" else T.gen_dna(N, 7)
text = np.concatenate([data, np.zeros(1, dtype=np.uint8)])
with T.Context(0) as ctx:
    for _ in range(2):
        out, st = ctx.lcpcomp_compress(text, thr, 1, coder, comp)
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items() if k.startswith("ms_") or k.startswith("sa_") or k in ("out_len", "factors", "levels")})
