"""Development aid: eager runs of small levels on / off on a synthetic text -- same stream, stage times.
Usage: python3 tools/eager_check.py dna|english N [threshold] [arith|huff]"""
import os; os.environ.setdefault("TDC_GPU_DEBUG_KNOBS", "1")   # (development tool: the TDC_GPU_* variables below are applied -- include/tdc_gpu.h, options)
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tudocomp_amd as T

gen, N = sys.argv[1], int(float(sys.argv[2]))
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 2
coder = T.CODER_ARITH if (len(sys.argv) > 4 and sys.argv[4] == "arith") else T.CODER_HUFF
data = T.gen_english(N, 42) if gen == "english" else T.gen_dna(N, 7)
text = np.concatenate([data, np.zeros(1, dtype=np.uint8)])
res = {}
for mode in ("1", "0"):
    os.environ["TDC_GPU_EAGER"] = mode
    with T.Context(0) as ctx:
        for _ in range(2):
            out, st = ctx.lcpcomp_compress(text, thr, 1, coder)
        res[mode] = hashlib.sha256(bytes(out)).hexdigest()
        print("eager", mode, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in st.items()
                              if k.startswith("ms_") or k in ("out_len", "factors", "levels", "small_levels", "eager_levels", "eager_phases", "pushes", "window_pass")})
print("same stream:", res["1"] == res["0"])
