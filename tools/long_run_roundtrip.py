import sys, time
sys.path.insert(0, "/root/repo")
import tudocomp_amd as T
from oracle import oracle as O
for N in (4_000_000, 16_000_000):
    for name, data in (("a^N", b"a" * N), ("(ab)^N/2", b"ab" * (N // 2))):
        text = O.escape(data)
        with T.Context(0) as ctx:
            t0 = time.time(); out, st = ctx.lzss_lcp_compress(text, 3); t1 = time.time()
            back, ds = ctx.lcpcomp_decompress(out); t2 = time.time()
            print(N, name, "lzss_lcp %.3f s" % (t1 - t0), "decompress %.3f s rounds %d" % (t2 - t1, ds["rounds"]), back == text, flush=True)
            t0 = time.time(); out, st = ctx.lcpcomp_compress(text, 5, 1); t1 = time.time()
            back, ds = ctx.lcpcomp_decompress(out); t2 = time.time()
            print(N, name, "lcpcomp  %.3f s" % (t1 - t0), "decompress %.3f s rounds %d" % (t2 - t1, ds["rounds"]), back == text, flush=True)
