"""Decompression timing: tdc_gpu_lcpcomp_decompress on the stream of a synthetic text, token stream parsed on the
host (TDC_GPU_DEC_PARSE=0) vs on the device (default).  Usage: python3 tools/decode_bench.py [english|dna] [N] [threshold]"""
import os; os.environ.setdefault("TDC_GPU_DEBUG_KNOBS", "1")   # (development tool: the TDC_GPU_* variables below are applied -- include/tdc_gpu.h, options)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tudocomp_amd as T

gen = sys.argv[1] if len(sys.argv) > 1 else "english"
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1 << 28
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 2
data = T.gen_english(N, 42) if gen == "english" else T.gen_dna(N, 7)
text = np.concatenate([data, np.zeros(1, dtype=np.uint8)])
with T.Context(0) as ctx:
    stream, cst = ctx.lcpcomp_compress(text, thr, 1)
print("%s %d B, threshold %d: stream %d B, %d factors, fdist_max %d" % (gen, N, thr, len(stream), cst["factors"], cst["fdist_max"]), flush=True)
want = text.tobytes()
for mode in ("0", "1"):
    os.environ["TDC_GPU_DEC_PARSE"] = mode
    with T.Context(0) as ctx:
        import ctypes
        ts = []
        a = np.frombuffer(stream, dtype=np.uint8)
        for i in range(4):                         # the C ABI call alone (the binding's copy into a Python bytes object is not the library's time)
            p, n = ctypes.c_void_p(), ctypes.c_size_t()
            t0 = time.perf_counter()
            rc = ctx._L.tdc_gpu_lcpcomp_decompress_coder(ctx._h, a.ctypes.data_as(ctypes.c_void_p), len(a), T.CODER_HUFF, ctypes.byref(p), ctypes.byref(n), None, None)
            ts.append(time.perf_counter() - t0)
            assert rc == 0
            ctx._L.tdc_gpu_free(p)
        back, st = ctx.lcpcomp_decompress(stream)
        ok = back == want
        del back
        t = min(ts[1:])
        print("TDC_GPU_DEC_PARSE=%s: device_parse %d, rounds %d, best of 3 %.1f ms = %.2f GB/s of text, correct %s (all: %s)"
              % (mode, st["device_parse"], st["rounds"], t * 1e3, N / 1e9 / t, ok, " ".join("%.1f" % (x * 1e3) for x in ts)), flush=True)

# the same call into a caller-owned pinned buffer (tdc_gpu_lcpcomp_decompress_into), stream in pinned memory as well
os.environ["TDC_GPU_DEC_PARSE"] = "1"
with T.Context(0) as ctx:
    h_in = T.PinnedBuffer(len(stream)); h_in.a[:] = np.frombuffer(stream, dtype=np.uint8)
    h_out = T.PinnedBuffer(N + 1)
    ts = []
    for i in range(5):
        t0 = time.perf_counter()
        n, st = ctx.lcpcomp_decompress_into(h_in, h_out)
        ts.append(time.perf_counter() - t0)
    ok = n == N + 1 and h_out.a[:n].tobytes() == want
    t = min(ts[1:])
    print("decompress_into (pinned buffers): device_parse %d, best of 4 %.1f ms = %.2f GB/s of text, correct %s (all: %s)"
          % (st["device_parse"], t * 1e3, N / 1e9 / t, ok, " ".join("%.1f" % (x * 1e3) for x in ts)), flush=True)
    h_in.free(); h_out.free()
