"""Development aid: one degenerate text (a^N | (ab)^N/2 | x a^N | a^N/2 x a^N/2) through lcpcomp(threshold=5), stage times or the error.
Usage: python3 tools/long_run_probe.py KIND N [comp=arrays|max_lcp|plcppeaks|heap] [option=value ...]   (with TDC_GPU_LIB pointing at a variant built with a back trace in
Arena::alloc this shows which allocation an out-of-memory error came from)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tudocomp_amd as T
from oracle import oracle as O
kind, N = sys.argv[1], int(float(sys.argv[2]))
opts = dict(a.split("=") for a in sys.argv[3:])
comp = {"arrays": T.COMP_ARRAYS, "max_lcp": T.COMP_MAXLCP, "plcppeaks": T.COMP_PLCPPEAKS, "heap": T.COMP_HEAP}[opts.pop("comp", "arrays")]
data = {"a": b"a" * N, "ab": b"ab" * (N // 2), "xa": b"x" + b"a" * (N - 1), "axa": b"a" * (N // 2) + b"x" + b"a" * (N // 2)}[kind]
text = O.escape(data)
with T.Context(0, options=opts) as ctx:
    try:
        out, st = ctx.lcpcomp_compress(text, 5, 1, T.CODER_HUFF, comp)
        print(kind, N, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in st.items() if k.startswith("ms_") or k in ("arena_bytes", "sa_rounds", "levels", "sa_mode")}, "out", len(out), flush=True)
        back, _ = ctx.lcpcomp_decompress(out)
        print("round trip", back == text)
    except T.TdcGpuError as e:
        print(kind, N, "FAILED:", e, flush=True)
