"""LZ78 + Elias-gamma (BASELINE configs[3]) on a GPU box: throughput of the host parse + device packing, and a 4 MiB prefix against the oracle.
Usage: python3 tools/lz78_check.py [bytes]"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tudocomp_amd as T
from oracle import oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 26)
d = T.gen_english(N, 42)
with T.Context(0) as ctx:
    t=time.time(); out, st = ctx.lz78_compress(d); dt=time.time()-t
    print("lz78 %d B: %.2f s = %.1f MB/s, factors %d" % (N, dt, N/1e6/dt, st["factors"]))
    want = O.lz78_gamma_compress(d[:1<<22]); got,_ = ctx.lz78_compress(d[:1<<22]); print("equal", got==want)
