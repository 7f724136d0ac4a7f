"""Stage times on texts that are one long run / one short period (PLCP = n - i - 1: the worst case for restarts of the
Phi algorithm and for the number of LCP levels).  Run on a GPU box: python tools/long_run_check.py N [N ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tudocomp_amd as T
from oracle import oracle as O

for N in [int(float(a)) for a in sys.argv[1:]] or [4_000_000]:
    for name, data in (("a^N", b"a" * N), ("(ab)^N/2", b"ab" * (N // 2)), ("x a^N (a jump right behind position 0)", b"x" + b"a" * (N - 1)),
                       ("a^N/2 x a^N/2", b"a" * (N // 2) + b"x" + b"a" * (N // 2))):
        text = O.escape(data)
        with T.Context(0) as ctx:
            try:
                for _ in range(2):
                    out, st = ctx.lcpcomp_compress(text, 5, 1)
                print(N, name, {k: round(v, 1) for k, v in st.items() if k.startswith("ms_")}, "out", len(out), flush=True)
            except T.TdcGpuError as e:
                print(N, name, "FAILED:", e, flush=True)
