"""Kernel-class times of one compression of the 10^9 B DNA text (BASELINE configs[2]) with the arithmetic coder (development aid).
Usage: python3 tools/dna_profile.py [option=value ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tudocomp_amd as T
opts = dict(a.split("=") for a in sys.argv[1:])
n = 1_000_000_000
text = np.concatenate([T.gen_dna(n, 7), np.zeros(1, dtype=np.uint8)])
with T.Context(0, options=opts) as ctx:
    ctx.lcpcomp_compress(text, 2, 1, T.CODER_ARITH)
    ctx.set_profiling(True); ctx.reset_profile()
    out, st = ctx.lcpcomp_compress(text, 2, 1, T.CODER_ARITH)
    print(opts, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in st.items() if k.startswith("ms_") or k in ("sa_rounds", "sa_star_chains", "out_len", "levels")})
    kp = ctx.kernel_profile()
    print(" ".join("%s=%.1f(%d)" % (k, v["ms"], v["launches"]) for k, v in sorted(kp.items(), key=lambda x: -x[1]["ms"]) if v["ms"] > 0.3))
